"""Host restatement of what ygz_hip_ba_build_windows assembles for one BA window -- the tests compare the device-built graph with it
(tests/test_gpu_offline.py::test_device_built_window_equals_host_built).  Test infrastructure: it lived in ygz_slam_amd/offline.py until round 6."""
import numpy as np
from ygz_slam_amd.offline import I7, se3_act, se3_mul, se3_log_g2o


def build_window_host(kf_tab, kfs, T_rel, fx, fy, cx, cy, max_points, match_sets=None, direct=None, width=0, height=0):
    """Host restatement of what ygz_hip_ba_build_windows assembles for one window (the tests compare the device-built graph with it):
    kf_tab[f] = dict(px, level, desc, depth) of keyframe f, match_sets(descs, pair_q, pair_t) = HipContext.match_sets.  Returns the
    graph of ba::LocalBAG2O in the anchor's gauge: poses (g2o order), points, edges sorted by (point, keyframe).
    With direct = fn(ref_frame, cur_frame, T_cur, px_ref, depth_ref, level_ref, px_cur) -> (ok, px) (a per-pair FindDirectProjection, e.g.
    HipContext.find_direct_projection on a context that holds the keyframes' pyramids) the observations are those of obs_mode 1: the map
    point projected with the chained pose, FindCandidates' test (z >= 0, InFrame(px, 20) of a width x height frame), FindDirectProjection."""
    A = kf_tab[kfs[0]]
    sel = np.nonzero(A["depth"] > 0)[0][:max_points]
    z = A["depth"][sel]
    pc = np.stack([(A["px"][sel, 0] - cx) * z / fx, (A["px"][sel, 1] - cy) * z / fy, z], axis=1)     # Pixel2Camera (Camera.h:56-62)
    ep, el, obs = [np.zeros(len(sel), np.int32)], [np.arange(len(sel), dtype=np.int32)], [A["px"][sel]]
    others = [(j, f) for j, f in enumerate(kfs[1:], start=1)]
    if others and len(sel) and direct is not None:
        from ygz_slam_amd import _lib
        Tc = _lib.se3_chain(T_rel[kfs[0]:kfs[-1] + 1])            # T(anchor) = identity, the same Sophus products as the device's chain
        for j, f in others:
            T = Tc[f - kfs[0]]
            q = se3_act(T, pc)
            pred = np.stack([fx * q[:, 0] / q[:, 2] + cx, fy * q[:, 1] / q[:, 2] + cy], axis=1)            # Camera2Pixel (Camera.h:46-51)
            vis = ~(q[:, 2] < 0) & (pred[:, 0] >= 20) & (pred[:, 0] < width - 20) & (pred[:, 1] >= 20) & (pred[:, 1] < height - 20)
            g = np.nonzero(vis)[0]
            if len(g):
                ok, pxo = direct(kfs[0], f, T, A["px"][sel][g], z[g], A["level"][sel][g], pred[g])
                g = g[ok]; pxo = pxo[ok]
            else:
                pxo = np.zeros((0, 2))
            ep.append(np.full(len(g), j, np.int32)); el.append(g.astype(np.int32)); obs.append(pxo)
    elif others and len(sel):
        res = match_sets([A["desc"][sel]] + [kf_tab[f]["desc"] for _, f in others], [0] * len(others), list(range(1, len(others) + 1)))
        for (j, f), r in zip(others, res):
            g = np.nonzero(r["good"])[0]
            ep.append(np.full(len(g), j, np.int32)); el.append(g.astype(np.int32)); obs.append(kf_tab[f]["px"][r["idx"][g]])
    ep, el, obs = np.concatenate(ep), np.concatenate(el), np.concatenate(obs)
    n_obs = np.bincount(el, minlength=len(sel))
    keep_pt = n_obs >= 2                                   # a point seen only by the fixed anchor constrains nothing
    remap = -np.ones(len(sel), np.int64); remap[keep_pt] = np.arange(int(keep_pt.sum()))
    ke = keep_pt[el]
    ep, el, obs = ep[ke], remap[el[ke]].astype(np.int32), obs[ke]
    order = np.lexsort((ep, el))
    T = I7.copy()
    poses = [se3_log_g2o(T)]
    for f in range(kfs[0] + 1, kfs[-1] + 1):
        T = se3_mul(T_rel[f], T)
        if f in kfs:
            poses.append(se3_log_g2o(T))
    fixed = np.zeros(len(kfs), np.uint8); fixed[0] = 1
    return dict(kfs=list(kfs), poses=np.stack(poses), fixed=fixed, points=pc[keep_pt], edge_pose=ep[order], edge_point=el[order], obs=obs[order],
                anchor_feature=sel[keep_pt])
