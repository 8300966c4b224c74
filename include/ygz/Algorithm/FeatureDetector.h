// ygz::FeatureDetector -- same surface as include/ygz/Algorithm/FeatureDetector.h:42-99; the work runs in
// libygz_hip.so (ygz_hip_detect / ygz_hip_describe).
#ifndef YGZ_FEATUREDETECTOR_H_
#define YGZ_FEATUREDETECTOR_H_
#include "ygz/Basic/Common.h"
#include "ygz/Basic/Frame.h"
namespace ygz {
class FeatureDetector {
public:
    const int PATCH_SIZE = 31;
    const int HALF_PATCH_SIZE = 15;
    const int EDGE_THRESHOLD = 19;
    struct Option {
        int _image_width = 640, _image_height = 480;
        int _cell_size = 10;                 // the reference leaves this uninitialised until LoadParams()
        int _grid_rows = 0, _grid_cols = 0;
        double _detection_threshold = 20.0;
    } _option;
    FeatureDetector();
    void LoadParams();
    void Detect(Frame *frame, bool overwrite_existing_features = true);
    void ComputeAngleAndDescriptor(Frame *frame);
    void ComputeDescriptor(Feature *fea);
private:
    void SetExistingFeatures(Frame *frame);
    vector<Feature *> _old_features;
    vector<Feature *> _new_features;
};
}
#endif
