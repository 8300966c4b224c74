#!/usr/bin/env python3
"""Generate tests/golden/*.npz: seeded inputs + the ORACLE's outputs for every stage.

The reference holds no golden vectors and cannot be built, so these fixtures are
produced by oracle/ (our CPU restatement).  They (a) pin the oracle against accidental change and
(b) are what the GPU parity tests are checked against besides the live oracle.  Known-answer
content that does come from the reference: the 8 poses / 16 points of test/test_local_ba.cpp:9-37
(zero-noise residuals must vanish).  Run from the repo root: python tools/make_golden.py
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Oracle
from ygz_slam_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fixtures

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
o = Oracle()
rng = np.random.default_rng(2024)


def small_frames(n=2, w=320, h=240, seed=3, step=0.6):
    tex, m = synth.make_texture(seed, w, h, margin=80)
    poses = synth.trajectory(n, seed + 10, step)
    poses[0] = [0, 0, 0, 1, 0, 0, 0]      # reference frame at the origin (GetWarpAffineMatrix mixes frames otherwise)
    imgs, deps = [], []
    for i in range(n):
        im, d = synth.render(tex, m, poses[i], w, h, 1.0, seed * 100 + i)
        imgs.append(im); deps.append(d)
    return np.stack(imgs), poses, np.stack(deps)


# 1 image ops
bgr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
img = rng.integers(0, 256, (45, 67), dtype=np.uint8)
np.savez_compressed(os.path.join(OUT, "image.npz"), bgr=bgr, gray=o.bgr2gray(bgr), img=img,
                    l1=o.pyr_down(img), l2=o.pyr_down(o.pyr_down(img)))

# 2 FAST + extractor on a 320x240 frame pair
imgs, poses, deps = small_frames()
lv0 = o.pyramid(imgs[0], 3)
fx = {}
for L in range(3):
    xy = o.fast_detect(lv0[L], 15)
    sc = o.fast_score(lv0[L], xy, 15)
    nm = o.fast_nonmax(xy, sc, 0)
    nm1 = o.fast_nonmax(xy, sc, 1)
    fx["xy%d" % L], fx["sc%d" % L], fx["nm%d" % L], fx["nmtie%d" % L] = xy, sc, nm, nm1
prm = o.default_params(320, 240, 3)
k0 = o.detect(lv0, prm)
k1 = o.detect(o.pyramid(imgs[1], 3), prm)
occ = np.zeros(32 * 24, np.uint8); occ[::3] = 1
k0occ = o.detect(lv0, prm, occ)
np.savez_compressed(os.path.join(OUT, "extract.npz"), imgs=imgs, poses=poses, k0=k0, k1=k1, occ=occ, k0occ=k0occ, **fx)

# 3 Hamming
q = fixtures.random_descriptors(70, 5); t = fixtures.random_descriptors(53, 6)
t[10] = q[3]; t[11] = q[3]; q[20] = q[21]         # exact ties
hm = dict(q=q, t=t)
for cc in (0, 1, 2):
    idx, d, n = o.bf_match(q, t, cc)
    hm["idx%d" % cc], hm["dist%d" % cc] = idx, d
idx, d, d2 = o.hamming_nn(q, t)
hm["nn_idx"], hm["nn_d"], hm["nn_d2"] = idx, d, d2
i2, dd2, _ = o.bf_match(k0["desc"], k1["desc"], 1)
hm["kidx"], hm["kdist"] = i2, dd2
np.savez_compressed(os.path.join(OUT, "hamming.npz"), **hm)

# 4 align2d / direct projection / sparse align / klt on the pair
n = min(200, len(k0))
sel = k0[:n]
px_ref = np.stack([sel["px"], sel["py"]], 1)
depth = np.array([deps[0][int(p[1]), int(p[0])] for p in px_ref])
T_ref, T_cur = poses[0], poses[1]
lv1 = o.pyramid(imgs[1], 3)
# predicted pixel = true projection + deterministic offset in [-2,2]
R = synth.quat_to_R(o.se3_mul(T_cur, o.se3_inv(T_ref))[:4]); tt = o.se3_mul(T_cur, o.se3_inv(T_ref))[4:]
pc = np.stack([(px_ref[:, 0] - synth.CX) / synth.FX * depth, (px_ref[:, 1] - synth.CY) / synth.FY * depth, depth], 1) @ R.T + tt
pred = np.stack([synth.FX * pc[:, 0] / pc[:, 2] + synth.CX, synth.FY * pc[:, 1] / pc[:, 2] + synth.CY], 1)
pred += rng.uniform(-2, 2, pred.shape)
ok, pxo, sl = [], [], []
for i in range(n):
    a, b, c = o.find_direct_projection(lv0, T_ref, lv1, T_cur, px_ref[i], depth[i], int(sel["level"][i]), pred[i])
    ok.append(a); pxo.append(b); sl.append(c)
has_mp = np.ones(n, np.uint8); has_mp[::7] = 0
T_init = o.se3_mul(synth.se3_exp([0.004, -0.003, 0.002, 0.001, -0.002, 0.001]), T_cur)
nmeas, T_est, st = o.sparse_align(lv0, T_ref, lv1, T_init, px_ref, depth, has_mp)
pts0 = px_ref.astype(np.float32)
init = (pred + 3.0).astype(np.float32)
kp, kst, kerr = o.klt_track(imgs[0], imgs[1], pts0, init)
np.savez_compressed(os.path.join(OUT, "align.npz"), px_ref=px_ref, depth=depth, level=sel["level"], pred=pred,
                    fdp_ok=np.array(ok), fdp_px=np.array(pxo), fdp_sl=np.array(sl), has_mp=has_mp, T_init=T_init,
                    sa_nmeas=nmeas, sa_T=T_est, sa_iters=np.array(list(st.iters_per_level)[:3]),
                    klt_init=init, klt_pts=kp, klt_status=kst, klt_err=kerr)

# 5 BA: transcription of test/test_local_ba.cpp (zero-noise = known answer) + noisy + outliers
for name, noise in (("ba_exact", False), ("ba_noisy", True)):
    f = fixtures.ba_fixture_test_local_ba(noise=noise)
    if noise:
        f["obs"][5] += 40.0          # beyond the Huber delta
    r = o.ba_linearize(f["poses"], f["fixed"], f["points"], f["edge_pose"], f["edge_point"], f["obs"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: v for k, v in f.items()},
                        **{"o_" + k: np.asarray(v) for k, v in r.items()})
print("golden fixtures written to", OUT, [(f, os.path.getsize(os.path.join(OUT, f))) for f in sorted(os.listdir(OUT))])
