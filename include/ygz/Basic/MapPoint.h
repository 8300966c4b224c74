// ygz::MapPoint -- same fields as include/ygz/Basic/MapPoint.h:17-46.
#ifndef YGZ_MAP_POINT_H_
#define YGZ_MAP_POINT_H_
#include "ygz/Basic/Common.h"
namespace ygz {
struct Feature;
struct MapPoint {
    MapPoint() {}
    inline float GetFoundRatio() const { return (float)_cnt_found / _cnt_visible; }
    unsigned long _id = 0;
    Vector3d      _pos_world = Vector3d(0, 0, 0);
    map<unsigned long, Feature *> _obs;
    bool          _bad = false;
    Mat           _distinctive_desc;
    unsigned long _first_seen = 0, _last_seen = 0;
    int           _cnt_visible = 0, _cnt_found = 0;
    bool          _track_in_view = false;
};
}
#endif
