"""CPU checks of the BoW oracle (oracle/bow.c: DBoW3 tree descent, BowVector/FeatureVector, Matcher::SearchByBoW,
Matcher::SearchForTriangulation) against straightforward numpy re-derivations on a synthetic vocabulary in DBoW3's binary
format (the reference does not ship vocab/ORBvoc.bin)."""
import struct
import numpy as np
from ygz_slam_amd import synth
import fixtures

LUT = np.array([bin(i).count("1") for i in range(256)])


def _parse(blob):
    nb, sz, k, L, sc, wt = struct.unpack_from("<IIiiii", blob, 0)
    nodes = [dict(parent=-1, desc=np.zeros(32, np.uint8), w=0.0, leaf=0, children=[])]
    for i in range(nb):
        off = 24 + i * sz
        par, = struct.unpack_from("<i", blob, off)
        d = np.frombuffer(blob, np.uint8, 32, off + 4).copy()
        w, = struct.unpack_from("<f", blob, off + 36)
        nodes.append(dict(parent=par, desc=d, w=float(w), leaf=blob[off + 40], children=[]))
        nodes[par]["children"].append(i + 1)
    wid = 0
    for nd in nodes[1:]:
        if nd["leaf"]:
            nd["word"] = wid; wid += 1
    return k, L, nodes


def _descend(nodes, L, d, levelsup):
    cur, level, nid = 0, 0, 0
    while nodes[cur]["children"]:
        level += 1
        ch = nodes[cur]["children"]
        dist = [int(LUT[d ^ nodes[c]["desc"]].sum()) for c in ch]
        cur = ch[int(np.argmin(dist))]                    # first minimum
        if level == L - levelsup:
            nid = cur
    return nodes[cur]["word"], nodes[cur]["w"], nid


def test_vocab_parse_and_transform(oracle):
    blob = fixtures.synthetic_vocabulary(k=7, L=4, seed=2)
    k, L, nodes = _parse(blob)
    v = oracle.vocab_parse(blob)
    assert (v.k, v.L, v.n_nodes) == (7, 4, len(nodes)) and v.n_words == 7 ** 4
    desc = fixtures.random_descriptors(300, 8)
    for levelsup in (0, 1, 2, 4, 6):
        word, weight, node, bw, bv = oracle.bow_transform(v, desc, levelsup)
        acc = {}
        for i in range(len(desc)):
            w, wt, nid = _descend(nodes, L, desc[i], levelsup)
            assert (word[i], weight[i]) == (w, wt), i
            assert node[i] == (nid if wt > 0 else -1), (i, levelsup)
            if wt > 0:
                acc[w] = acc.get(w, 0.0) + wt
        ks = sorted(acc)
        assert list(bw) == ks
        tot = sum(abs(acc[q]) for q in ks)
        assert np.allclose(bv, [acc[q] / tot for q in ks], rtol=1e-15) and abs(bv.sum() - 1) < 1e-12
    assert (oracle.bow_transform(v, desc, 1)[2] < 0).sum() > 0              # some words are stopped
    # malformed blobs are rejected
    import pytest
    with pytest.raises(ValueError):
        oracle.vocab_parse(blob[:100])


def test_search_by_bow_and_triangulation(oracle):
    blob = fixtures.synthetic_vocabulary(k=10, L=3, seed=5)
    v = oracle.vocab_parse(blob)
    rng = np.random.default_rng(3)
    d1 = fixtures.random_descriptors(400, 11)
    d2 = d1[rng.permutation(400)].copy()
    flip = rng.random((400, 256)) < 0.04
    d2 ^= np.packbits(flip, axis=1)
    d2 = np.concatenate([d2, fixtures.random_descriptors(100, 12)])
    n1 = oracle.bow_transform(v, d1, 1)[2]; n2 = oracle.bow_transform(v, d2, 1)[2]
    m, cnt = oracle.search_by_bow(d1, n1, d2, n2, th_low=65, knn_ratio=0.7)
    D = LUT[d1[:, None, :] ^ d2[None, :, :]].sum(-1)
    exp = np.full(400, -1)
    for i in range(400):
        if n1[i] < 0:
            continue
        js = np.nonzero(n2 == n1[i])[0]
        if len(js) == 0:
            continue
        dd = D[i, js]
        o = np.argsort(dd, kind="stable")
        b1 = dd[o[0]]; b2 = dd[o[1]] if len(js) > 1 else 256
        if b1 < 65 and np.float32(b1) < np.float32(0.7) * np.float32(b2):
            exp[i] = js[o[0]]
    assert np.array_equal(m, exp) and cnt == (exp >= 0).sum() and cnt > 100
    # triangulation: pixels related by a pure x-translation -> E = [t]x, epipolar lines are the image rows
    px1 = np.stack([rng.uniform(50, 590, 400), rng.uniform(50, 430, 400)], 1)
    inv = np.empty(500, int); inv[:] = -1
    px2 = np.stack([rng.uniform(50, 590, 500), rng.uniform(50, 430, 500)], 1)
    E = np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
    mt, ct = oracle.search_for_triangulation(d1, n1, px1, d2, n2, px2, E, th_low=65, epipolar_dsqr=1e-4)
    cam = oracle.camera()
    expt = np.full(400, -1)
    for i in range(400):
        if n1[i] < 0:
            continue
        best, bi = 256, -1
        y1 = (px1[i, 1] - np.float64(cam.cy)) / np.float64(cam.fy)
        for j in np.nonzero(n2 == n1[i])[0]:
            if D[i, j] > 65 or D[i, j] > best:
                continue
            y2 = (px2[j, 1] - np.float64(cam.cy)) / np.float64(cam.fy)
            # a = 0, b = -1 (E[1][2])... evaluate with the same float steps
            x1 = (px1[i, 0] - np.float64(cam.cx)) / np.float64(cam.fx); x2 = (px2[j, 0] - np.float64(cam.cx)) / np.float64(cam.fx)
            a = np.float32(x1 * E[0, 0] + y1 * E[1, 0] + E[2, 0]); b = np.float32(x1 * E[0, 1] + y1 * E[1, 1] + E[2, 1]); c = np.float32(x1 * E[0, 2] + y1 * E[1, 2] + E[2, 2])
            num = np.float32(np.float64(a) * x2 + np.float64(b) * y2 + np.float64(c)); den = np.float32(a * a + b * b)
            if np.float64(den) < 1e-6:
                continue
            if np.float64(abs(np.float32(num * num / den))) < 1e-4:
                best, bi = D[i, j], j
        expt[i] = bi
    assert np.array_equal(mt, expt) and ct == (expt >= 0).sum()
    # empty second frame / nothing in common
    m0, c0 = oracle.search_by_bow(d1, n1, np.zeros((0, 32), np.uint8), np.zeros(0, np.int32))
    assert c0 == 0 and np.all(m0 == -1)


def test_bow_orientation_histogram(oracle):
    """the checkOrientation part of Matcher::SearchByBoW (Matcher.cpp:247-256, 271-289 + ComputeThreeMaxima :293-336) against a numpy
    re-derivation: float rot, bin = round(rot / 30) (the reference's factor: bins 0 .. 12 only), three maxima with the 0.1 rule, count left"""
    rng = np.random.default_rng(4)
    for trial in range(40):
        n1, n2 = int(rng.integers(1, 400)), int(rng.integers(1, 400))
        a1 = rng.uniform(0, 360, n1).astype(np.float32).astype(np.float64)
        a2 = rng.uniform(0, 360, n2).astype(np.float32).astype(np.float64)
        if trial % 3 == 0:                                      # a dominant rotation, as between two real frames
            m = rng.integers(0, n2, n1)
            a1 = (a2[m] + rng.normal(25, 4, n1)) % 360
        m = rng.integers(-1, n2, n1).astype(np.int32)
        if trial == 7:
            m[:] = -1
        cnt, hist, ind = oracle.bow_orientation(a1, a2, m)
        rot = (a1 - a2[np.maximum(m, 0)]).astype(np.float32)
        rot = np.where(rot < 0, rot + np.float32(360), rot).astype(np.float32)
        x = (rot * np.float32(1.0 / 30)).astype(np.float32).astype(np.float64)
        b = (np.sign(x) * np.floor(np.abs(x) + 0.5)).astype(int)          # C round(): half away from zero
        b[b == 30] = 0
        h = np.bincount(b[m >= 0], minlength=30)
        assert np.array_equal(hist, h) and h[13:].sum() == 0
        order = []                                                         # the scan of ComputeThreeMaxima, restated
        mx = [0, 0, 0]; ix = [-1, -1, -1]
        for i in range(30):
            s_ = h[i]
            if s_ > mx[0]:
                mx = [s_, mx[0], mx[1]]; ix = [i, ix[0], ix[1]]
            elif s_ > mx[1]:
                mx = [mx[0], s_, mx[1]]; ix = [ix[0], i, ix[1]]
            elif s_ > mx[2]:
                mx[2] = s_; ix[2] = i
        if np.float32(mx[1]) < np.float32(0.1) * np.float32(mx[0]):
            ix[1] = ix[2] = -1
        elif np.float32(mx[2]) < np.float32(0.1) * np.float32(mx[0]):
            ix[2] = -1
        assert list(ind) == ix
        assert cnt == int(sum(h[i] for i in set(ix) if i >= 0))
