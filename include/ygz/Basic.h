#ifndef YGZ_BASIC_H_
#define YGZ_BASIC_H_
#include "ygz/Basic/Common.h"
#include "ygz/Basic/Config.h"
#include "ygz/Basic/Camera.h"
#include "ygz/Basic/Feature.h"
#include "ygz/Basic/MapPoint.h"
#include "ygz/Basic/Frame.h"
#include "ygz/Basic/Memory.h"
#endif
