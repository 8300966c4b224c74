// ygz::Config -- key/value parameters (reference: include/ygz/Basic/Config.h:14-40, src/Basic/Config.cpp:6-18).
// The reference reads config/default.yaml through cv::FileStorage; here a tiny "key: value" reader with the
// defaults of config/default.yaml:8-66 built in, same Get<T>(key) interface.
#ifndef YGZ_CONFIG_H_
#define YGZ_CONFIG_H_
#include "ygz/Basic/Common.h"
namespace ygz {
class Config {
public:
    static bool SetParameterFile(const std::string &filename);     // parses "key: value" lines, '#'/'%' comments
    static void Set(const std::string &key, const std::string &value);
    template <typename T> static T Get(const std::string &key)
    {
        std::istringstream is(Raw(key));
        double v = 0; is >> v;                 // every numeric key of default.yaml; absent keys (camera.k1..p2) read 0
        return static_cast<T>(v);
    }
    static std::string Raw(const std::string &key);
};
}
#endif
