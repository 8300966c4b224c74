"""Register budgets of the hot kernels, checked where they are decided: at compile time (hipcc cross-compiles gfx950 without a GPU).
The step's kernels are tuned to a number of wavefronts per SIMD; a change that pushes one of them over a
register boundary, or makes it spill to scratch memory, costs 5-20 % of its time and shows up in no parity test.  The numbers are
the compiler's own remarks (-Rpass-analysis=kernel-resource-usage) for the flags of ygz_slam_amd/csrc/Makefile."""
import concurrent.futures as cf
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ygz_slam_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel (substring of the mangled name) -> (file, minimum wavefronts per SIMD, maximum scratch bytes per lane)
BUDGET = {
    "k_klt3": ("klt", 4, 0),                       # 108 VGPRs: four wavefronts per SIMD (five needs 96 and spills)
    "k_scharr": ("klt", 8, 0),
    "k_klt_pad": ("klt", 8, 0),
    "k_hamming_f4ILi2E": ("hamming", 3, 0),        # the default matcher: 162 VGPRs, accumulators in VGPRs
    "k_sparse_align2ILi256EE": ("sparse_align", 1, 0),      # the batch form: 256 + <= 72 registers (more, and the matcher no longer fits beside it in the step)
    "k_sparse_align2ILi512EE": ("sparse_align", 2, 288),    # 720p problems and single-frame calls: capped at 256 registers per lane
    "k_fast_select": ("detect", 5, 0),            # two wavefronts per 64 x 32 tile, 15 KB of LDS: ten tiles per CU (measured faster than four wavefronts at eight per SIMD)
    "k_describeILi8E": ("detect", 7, 0),          # batches: eight keypoints per wavefront (the pattern pairs stay in registers)
    "k_describeILi1E": ("detect", 8, 0),          # single-frame calls
    "k_ba_points": ("ba", 3, 0),
    "k_find_direct_projection": ("align", 2, 0),
    "k_win_project": ("align", 2, 0),
    "k_pose_only_ba": ("pose_only", 2, 0),
    "k_ba_lm_team": ("ba_resident_lm", 1, 0),      # 256 lanes, one wavefront per SIMD, nothing spilled (it was 175 registers at 512 lanes)
    "k_bgr2gray16": ("image", 8, 0),
    "k_pyr_down": ("image", 8, 0),
}


def _usage(name):
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if name == "hamming" else []        # as in the Makefile
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", *extra,
                        "-I" + os.path.join(ROOT, "include"), "-c", name + ".hip", "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       cwd=CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_hot_kernels_keep_their_register_budget():
    files = sorted({f for f, _, _ in BUDGET.values()})
    with cf.ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        per_file = dict(zip(files, ex.map(_usage, files)))
    problems = []
    for key, (f, min_occ, max_scratch) in BUDGET.items():
        hits = [(k, v) for k, v in per_file[f].items() if key in k]
        assert len(hits) == 1, (key, [k for k, _ in hits])
        k, v = hits[0]
        if v["Occupancy"] < min_occ or v["ScratchSize"] > max_scratch:
            problems.append("%s: %d wavefronts per SIMD (budget %d), %d VGPRs + %d AGPRs, scratch %d B per lane (budget %d)"
                            % (k, v["Occupancy"], min_occ, v["VGPRs"], v["AGPRs"], v["ScratchSize"], max_scratch))
    # the resident sparse alignment shares every SIMD with the matcher (168 registers) and one LK wavefront: beyond 344 registers per lane
    # the matcher no longer fits beside it and the step gets 10 % slower (measured with 382)
    (k, v), = [(k, v) for k, v in per_file["sparse_align"].items() if "k_sparse_align2ILi256EE" in k]
    if v["VGPRs"] + v["AGPRs"] > 344:
        problems.append("%s: %d + %d registers per lane (budget 344)" % (k, v["VGPRs"], v["AGPRs"]))
    assert not problems, "\n".join(problems)
