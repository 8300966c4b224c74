#!/usr/bin/env python3
"""Static VALU instruction mix of the hot loops of k_klt3 and k_hamming_nn (no GPU needed: hipcc -S of the kernel sources).

The issue ceiling of a SIMD depends on the opcode class (profiles/r02_valu_peak.txt, measured on MI355X with 8 wavefronts per SIMD):
  full rate  v_add/sub/mul/fma_f32, v_xor/and/or, v_add/sub_u32 ...       0.93 G wave64-instr/s/SIMD  (2.6 cycles at 2.4 GHz)
  half rate  v_bcnt, v_perm, v_alignbyte, v_dot2, v_mad_*24, v_mul_lo, v_add3, v_lshl_add, v_bfe, v_cvt, v_sad, v_min/max_u32,
             v_lshlrev, DPP-modified ops, every FP64 op                    0.575 G wave64-instr/s/SIMD (4.2 cycles)
A kernel whose VALU instructions are a share h of half-rate ops cannot issue faster than 1 / (h / 0.575 + (1 - h) / 0.93).
Opcodes that were not measured are classed by family (documented in CLASS below) and listed in the output.
usage: tools/valu_mix.py  -> profiles/valu_mix.json"""
import collections, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL, HALF = 0.93, 0.575
HALF_PREFIX = ("v_bcnt", "v_perm", "v_alignb", "v_dot2", "v_dot4", "v_mad_i32_i24", "v_mad_u32_u24", "v_mul_i32_i24", "v_mul_u32_u24", "v_mul_lo", "v_mul_hi",
               "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_bfe", "v_bfi", "v_cvt", "v_sad", "v_min_u", "v_max_u", "v_min_i",
               "v_max_i", "v_min3", "v_max3", "v_med3", "v_lshlrev", "v_lshrrev", "v_ashrrev", "v_pk_", "v_readlane", "v_readfirstlane", "v_writelane",
               "v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_rndne", "v_floor", "v_ceil", "v_trunc", "v_fract", "v_ldexp", "v_frexp",
               "v_div_", "v_mad_u64", "v_mad_i64", "v_add_lshl", "v_sub_co", "v_add_co", "v_addc", "v_subb")
MEASURED = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_xor_b32", "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_min_u32", "v_bcnt_u32_b32", "v_perm_b32",
            "v_alignbyte_b32", "v_dot2_i32_i16", "v_dot2c_i32_i16", "v_mad_i32_i24", "v_mad_u32_u24", "v_mul_lo_u32", "v_add3_u32", "v_lshl_add_u32", "v_bfe_u32",
            "v_cvt_f32_i32", "v_sad_u8", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_pk_fma_f32"}


def klass(op, line):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if "_f64" in base or "_u64" in base or "_i64" in base or "_b64" in base:
        return "half"
    if "dpp" in op or "row_" in line or "quad_perm" in line or "sdwa" in op:
        return "half"
    if base.startswith(HALF_PREFIX):
        return "half"
    return "full"


def hot_mix(src, symbol):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "ygz_slam_amd", "csrc"), src, "-o", out], stderr=subprocess.DEVNULL)
        s = open(out).read()
    i = s.index(symbol + ":")
    body = s[i:s.index("s_endpgm", i)]
    cnt, unmeasured = collections.Counter(), collections.Counter()
    in_loop = False
    for ln in body.split("\n"):
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            in_loop = ("Loop" in t)                      # LLVM marks blocks "in Loop: Header=..." / "Parent Loop ..." / "=>This Inner Loop Header"
            continue
        if not in_loop or not t.startswith("v_"):
            continue
        op = t.split()[0]
        if op.startswith("v_cmp") or op.startswith("v_mov") or op.startswith("v_cndmask") or op.startswith("v_accvgpr") or op.startswith("v_nop"):
            cnt["full"] += 1
            continue
        k = klass(op, t)
        cnt[k] += 1
        if re.sub(r"_(e32|e64|sdwa|dpp)$", "", op) not in MEASURED:
            unmeasured[op] += 1
    n = cnt["full"] + cnt["half"]
    h = cnt["half"] / n
    return {"valu_in_loops": n, "half_rate": cnt["half"], "full_rate": cnt["full"], "half_rate_share": h,
            "issue_ceiling_Ginstr_per_s_per_SIMD": 1.0 / (h / HALF + (1 - h) / FULL), "classed_by_family_not_measured": dict(unmeasured.most_common(12))}


res = {"note": __doc__.split("usage")[0].strip(), "peak_full_rate": FULL, "peak_half_rate": HALF, "peak_guide_2cycles_2p4GHz": 1.2,
       "source": "profiles/r02_valu_peak.txt",
       "kernels": {"k_klt3": hot_mix(os.path.join(ROOT, "ygz_slam_amd", "csrc", "klt.hip"), "_Z6k_klt37KltArgs"),
                   "k_hamming_nn": hot_mix(os.path.join(ROOT, "ygz_slam_amd", "csrc", "hamming.hip"), "_Z12k_hamming_nnILb0ELi2EEv7HamArgs")}}
json.dump(res, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
for k, v in res["kernels"].items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "classed_by_family_not_measured"})
