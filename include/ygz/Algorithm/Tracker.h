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
    list<Feature *> _tracked_features;
    vector<cv::Point2f> _px_curr;
    TrackerStatusType _status = TrackerStatusType::TRACK_NOT_READY;
};
}
#endif
