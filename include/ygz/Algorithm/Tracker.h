// ygz::Tracker -- same surface as include/ygz/Algorithm/Tracker.h:10-73; TrackKLT runs ygz_hip_klt_track.
#ifndef YGZ_TRACKER_H_
#define YGZ_TRACKER_H_
#include "ygz/Basic.h"
namespace ygz {
class Tracker {
public:
    enum TrackerStatusType { TRACK_NOT_READY, TRACK_GOOD, TRACK_LOST };
    struct Option {
        int _min_feature_tracking = 50;
        double klt_win_size = 21.0;
        int klt_max_iter = 30;
        double klt_eps = 0.001;
    } _option;
    Tracker();
    void SetReference(Frame *ref);
    void Track(Frame *curr);
    float MeanDisparity() const;
    void GetTrackedPixel(vector<Feature *> &feature1, vector<Vector2d> &pixels2) const;
    TrackerStatusType Status() const { return _status; }
private:
    void TrackKLT();
    Frame *_ref = nullptr, *_curr = nullptr;
    // the live tracks as parallel arrays (the reference keeps a std::list<Feature*> next to a vector<cv::Point2f>, Tracker.h:68-69):
    // row i = reference feature, its pixel (what LK starts from in the reference image) and its current position
    struct Tracks { vector<Feature *> feature; vector<float> ref_px, cur_px; size_t size() const { return feature.size(); } } _tracks;
    TrackerStatusType _status = TrackerStatusType::TRACK_NOT_READY;
};
}
#endif
