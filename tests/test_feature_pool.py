"""ygz::Feature's class-level operator new / delete (include/ygz/Basic/Feature.h, ygz::pool in Common.h): header-only, so a plain g++ program on the
CPU can hold it to what the callers rely on -- distinct live objects, independent descriptor blocks, reuse after delete, frees from another thread."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "ygz/Basic/Feature.h"
#include <set>
#include <thread>
#include <vector>
#include <cstdio>
using namespace ygz;
int main()
{
    std::vector<Feature *> a;
    std::set<void *> objs, descs;
    for (int i = 0; i < 3000; ++i) {                       // more than several slabs
        Feature *f = new Feature(Vector2d(i, 2 * i), i & 3, 0.5 * i);
        f->_desc.data[0] = (uint8_t)i; f->_desc.data[31] = (uint8_t)(i >> 8);
        a.push_back(f); objs.insert(f); descs.insert(f->_desc.data);
    }
    if (objs.size() != 3000 || descs.size() != 3000) { printf("live objects / descriptor blocks overlap\n"); return 1; }
    for (int i = 0; i < 3000; ++i)
        if (a[i]->_pixel[0] != i || a[i]->_level != (i & 3) || a[i]->_desc.data[0] != (uint8_t)i || a[i]->_desc.data[31] != (uint8_t)(i >> 8) ||
            a[i]->_desc.rows != 1 || a[i]->_desc.cols != 32 || a[i]->_depth != -1 || a[i]->_mappoint != nullptr) { printf("object %d damaged\n", i); return 2; }
    // a descriptor shared with a Mat copy outlives its Feature (cv::Mat semantics: reference counted)
    Mat keep = a[7]->_desc;
    for (int i = 0; i < 1500; ++i) delete a[i];
    if (keep.data[0] != 7) { printf("shared descriptor block was recycled under a live Mat\n"); return 3; }
    std::vector<Feature *> b;
    int reused = 0;
    for (int i = 0; i < 1500; ++i) { Feature *f = new Feature(Vector2d(0, 0)); reused += objs.count(f) ? 1 : 0; b.push_back(f); }
    if (reused < 1400) { printf("deleted blocks were not reused (%d)\n", reused); return 4; }
    for (Feature *f : b) if (f->_desc.data == keep.data) { printf("live descriptor block handed out twice\n"); return 5; }
    // the other half is deleted by another thread, which then allocates from what it was given
    std::thread t([&] { for (int i = 1500; i < 3000; ++i) delete a[i]; for (int i = 0; i < 2000; ++i) delete new Feature(Vector2d(1, 1)); });
    t.join();
    for (Feature *f : b) delete f;
    delete static_cast<Feature *>(nullptr);
    printf("ok\n");
    return 0;
}
'''


def test_feature_block_pool(tmp_path):
    src = tmp_path / "pool.cpp"
    src.write_text(SRC)
    exe = tmp_path / "pool"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                        "-I" + os.path.join(ROOT, "include"), "-o", str(exe), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr[-3000:])
