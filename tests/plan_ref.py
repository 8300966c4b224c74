"""Python restatement of the chunk plan of the offline run (the round-4 interpreter code): test infrastructure.  The product is
chunk_schedule / chunk_plan in ygz_slam_amd/host/ygz_offline.cpp (ygz_offline_plan_range); tests/test_dist_gloo.py holds the two equal on
random shards."""


def chunk_schedule(first, last, chunk, ramp=True, kf_stride=0, ramp_from=4):
    """[first, last) cut into chunks of `chunk` frames, with a ramp at both ends (chunk / 4, chunk / 2, chunk ... chunk, chunk / 2, chunk / 4)
    when there is room: nothing overlaps the upload of the first chunk or the kernels of the last one, so those two are kept short.
    kf_stride > 0: the frames behind the shard's last keyframe (they complete no BA window) form a chunk of their own at the very end, so
    that every window is complete one chunk earlier and the last resident-LM launch runs beside that chunk instead of after it"""
    n = last - first
    sizes = []
    if ramp and chunk >= 64 and n >= ramp_from * chunk:
        head = [chunk // 4, chunk // 2]
        tail = [chunk // 2, chunk // 4]
        body = n - sum(head) - sum(tail)
        sizes = head + [chunk] * (body // chunk) + ([body % chunk] if body % chunk else []) + tail
    else:
        sizes = [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
    out, c0 = [], first
    for s_ in sizes:
        out.append((c0, c0 + s_)); c0 += s_
    assert c0 == last or n <= 0
    if kf_stride > 0 and out:
        a, b = out[-1]
        k_last = ((b - 1) // kf_stride) * kf_stride                # the last keyframe of the shard
        if a <= k_last and k_last + 1 < b and k_last + 1 > a:
            out[-1:] = [(a, k_last + 1), (k_last + 1, b)]
    return out




def chunk_plan(first, last, chunk, ramp, kf_stride, windows, defer, group=45):
    """The chunks of the shard [first, last) IN PROCESSING ORDER; a chunk is a tuple of frame ranges ((a, b), ...) -- normally one.  Every
    frame pair is solved from the identity, so the order is free; what it decides is when a BA window is complete (all frames from its anchor
    to its last keyframe tracked) and therefore where its resident-LM launch -- a latency chain of ~5 ms that uses a fraction of the GPU --
    falls.  The frames BEHIND a window's last keyframe (kf_stride - 1 of them, up to the next anchor) complete nothing: for the last `defer`
    windows that end inside the shard they are taken out of the main pass and processed at the very end, about `group` frames per chunk (a
    chunk of several short ranges: one small chunk per gap costs a pass of latency-bound kernels each), so that the last LM launch runs
    beside their uploads and kernels instead of after everything else; the last chunk of the main pass -- the LM waits for its kernels --
    can be cut to last_main frames (measured slower, off).  Cost: two more halo frames per deferred gap (the range after a gap and the gap itself each upload their
    predecessor once more)."""
    plain = [((a, b),) for a, b in chunk_schedule(first, last, chunk, ramp, kf_stride)]
    if defer <= 0:
        return plain
    inside = [w for w in windows if w[0] >= first and w[-1] < last]
    anchors = sorted(w[0] for w in windows)
    gaps = []
    for w in inside[-defer:]:
        nxt = [a for a in anchors if a > w[-1]]
        g0, g1 = w[-1] + 1, min(last, nxt[0] if nxt else last)
        if g1 > g0 and g0 > first:
            gaps.append((g0, g1))
    if not gaps:
        return plain
    main, a = [], first
    for g0, g1 in gaps:
        if g0 > a:
            main.append((a, g0))
        a = g1
    if a < last:
        main.append((a, last))
    # the main pass: the schedule of a shard of n_main frames (ramp at both ends), its intervals mapped back onto the ranges that are left
    n_main = sum(b - a for a, b in main)
    virt = chunk_schedule(0, n_main, chunk, ramp, 0, ramp_from=3)
    out = []
    for v0, v1 in virt:
        rs, pos = [], 0
        for a, b in main:
            lo, hi = max(v0, pos), min(v1, pos + (b - a))
            if hi > lo:
                rs.append((a + lo - pos, a + hi - pos))
            pos += b - a
        out.append(tuple(rs))
    # the deferred gaps: whole gaps (a split gap would need one more halo frame), in n_groups chunks of about `group` frames each
    tot = sum(b - a for a, b in gaps)
    n_groups = max(1, int(round(tot / float(max(1, group)))))
    per = -(-len(gaps) // n_groups)
    for k in range(0, len(gaps), per):
        ch = []
        for g0, g1 in gaps[k:k + per]:
            while g1 - g0 > chunk:                                 # (a gap longer than a chunk)
                out.append(((g0, g0 + chunk),)); g0 += chunk
            ch.append((g0, g1))
        while sum(b - a for a, b in ch) > chunk:                   # (never more than `chunk` frames per chunk)
            out.append((ch.pop(0),))
        out.append(tuple(ch))
    return out


# ---- the chunk plan of the C++ driver itself (ygz_offline_plan_range of libygz_host.so) through ctypes: what the tests above compare the interpreter form with.
# (Lived in ygz_slam_amd/offline.py until round 6; it is test plumbing, not part of the product package.)
import ctypes as C
import numpy as np


def _plan_range(first, last, chunk, ramp, kf_stride, windows, defer):
    from ygz_slam_amd import offline
    lib = offline.host_lib()
    wfl = np.ascontiguousarray([[w[0], w[-1]] for w in windows], np.int32).reshape(-1, 2)
    cap = max(64, 2 * (last - first) + 64)
    out = np.zeros((cap, 3), np.int32)
    n = C.c_int(0)
    rc = lib.ygz_offline_plan_range(int(first), int(last), int(chunk), int(bool(ramp)), int(kf_stride), wfl.ctypes.data_as(C.POINTER(C.c_int32)), len(wfl),
                                    int(defer), out.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError("ygz_offline_plan_range failed: %d" % rc)
    plan = {}
    for ci, a, b in out[:n.value].tolist():
        plan.setdefault(ci, []).append((a, b))
    return [tuple(plan[k]) for k in sorted(plan)]


def cpp_chunk_schedule(first, last, chunk, ramp=True, kf_stride=0):
    """[first, last) in chunks of `chunk` frames with short chunks at both ends; kf_stride > 0: the frames behind the last keyframe form the
    last chunk (ygz_offline.cpp: chunk_schedule)"""
    return [ch[0] for ch in _plan_range(first, last, chunk, ramp, kf_stride, [], 0)]


def cpp_chunk_plan(first, last, chunk, ramp, kf_stride, windows, defer):
    """the chunks of the shard [first, last) in processing order, a chunk = a tuple of frame ranges (ygz_offline.cpp: chunk_plan)"""
    return _plan_range(first, last, chunk, ramp, kf_stride, windows, defer)
