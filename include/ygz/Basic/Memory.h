// ygz::Memory -- owner of keyframes / map points (include/ygz/Basic/Memory.h:16-56, src/Basic/Memory.cpp:21-71);
// only what ba::LocalBAG2O and the tests need.
#ifndef YGZ_MEMORY_H_
#define YGZ_MEMORY_H_
#include "ygz/Basic/Common.h"
namespace ygz {
struct Frame;
struct MapPoint;
class Memory {
public:
    static Frame *RegisterKeyFrame(Frame *frame, bool overwrite = false);   // assigns _keyframe_id
    static MapPoint *RegisterMapPoint(MapPoint *mp);                        // assigns _id
    static MapPoint *CreateMapPoint();                                      // src/Basic/Memory.cpp:45-52
    static Frame *GetKeyFrame(const unsigned long &keyframe_id);
    static MapPoint *GetMapPoint(const unsigned long &id);
    static void Clean();                                                    // forgets (does not delete)
};
}
#endif
