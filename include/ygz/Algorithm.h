#ifndef YGZ_ALGORITHM_H_
#define YGZ_ALGORITHM_H_
#include "ygz/Algorithm/FeatureDetector.h"
#include "ygz/Algorithm/Matcher.h"
#include "ygz/Algorithm/Tracker.h"
#include "ygz/Algorithm/SparseImageAlign.h"
#include "ygz/Algorithm/CVUtils.h"
#include "ygz/Algorithm/BA.h"
#endif
