"""The seven other sorting modes of the PPLL resolve pass (SORTING_MODE_NAMES, src/Renderers/PPLL.hpp:32-50;
Data/Shaders/Renderers/PPLL/LinkedListSort.glsl:62-172,241-262, LinkedListQuicksort.glsl:30-141): option `sorting_mode`.

CPU: the oracle's restatements against plain Python -- every mode must leave the list in the order Python's sorted() gives and
blend ALL fragments (no early out), except the two places where the shaders themselves do not sort: the bitonic network on a
list whose length is no power of two, and nothing else (the quicksort stack of ceil(log2 MAX_NUM_FRAGS) * 2 + 4 entries is deep
enough for random lists).  GPU: the resolve kernel against the oracle on the same lists, byte for byte, and whole frames."""
import numpy as np
import pytest

from common import small_case
from oracle import lvo

MODES = ["Priority Queue", "Bubble Sort", "Insertion Sort", "Shell Sort", "Max Heap", "Bitonic Sort", "Quicksort", "Quicksort Hybrid"]


def random_lists(W, H, max_frags, seed, lengths=None):
    rng = np.random.default_rng(seed)
    c = small_case(width=W, height=H, transparent=True, ppll_max_num_frags=max_frags, background=(0.2, 0.4, 0.6, 1.0))
    P = c.oracle_params()
    pw, ph = c.padded()
    start = np.full(pw * ph, 0xFFFFFFFF, dtype=np.uint32)
    nodes, lists = [], {}
    for y in range(H):
        for x in range(W):
            k = int(rng.integers(0, max_frags + 4)) if lengths is None else int(lengths[(y * W + x) % len(lengths)])
            if k == 0:
                continue
            depth = rng.uniform(0.3, 1.3, k).astype(np.float32)
            if k > 3 and rng.uniform() < 0.4:
                depth[rng.integers(0, k, 3)] = depth[0]        # exact depth ties
            col = rng.integers(0, 1 << 32, k, dtype=np.uint64).astype(np.uint32)
            if rng.uniform() < 0.3:
                col &= np.uint32(0x3FFFFFFF)                   # faint layers: nothing saturates early
            nxt = 0xFFFFFFFF
            for j in range(k):
                nodes.append((int(col[j]), int(depth[j].view(np.uint32)), nxt))
                nxt = len(nodes) - 1
            start[lvo.ppll_addr(x, y, pw, P.ppllTileW, P.ppllTileH)] = nxt
            lists[(x, y)] = (col[::-1][:max_frags].copy(), depth[::-1][:max_frags].copy())   # what the resolve reads, in list order
    return c, P, np.asarray(nodes, dtype=np.uint32).reshape(-1, 3), start, lists


def blend_all(cols, bg):
    """blendFTB (LinkedListSort.glsl:45-59) + BACK_TO_FRONT_STRAIGHT_ALPHA over the clear colour, in float32 like the shader."""
    f = np.float32
    color = [f(0), f(0), f(0), f(0)]
    for c in cols:
        src = [f(f((int(c) >> s) & 0xFF) / f(255.0)) for s in (0, 8, 16, 24)]
        for k in range(3):
            color[k] = f(color[k] + f(f(f(1.0) - color[3]) * src[3]) * src[k])
        color[3] = f(color[3] + f(f(1.0) - color[3]) * src[3])
    a = color[3]
    if not a > 0:
        out = [f(b) for b in bg]
    else:
        out = [f(f(f(color[k] / a) * a) + f(f(bg[k]) * f(f(1.0) - a))) for k in range(3)] + [f(a + f(f(bg[3]) * f(f(1.0) - a)))]
    return [int(np.floor(np.clip(v, 0, 1) * f(255.0) + f(0.5))) for v in out]


def bitonic_as_written(keys):
    """The network of LinkedListSort.glsl:241-262 on a Python list of keys, with its guards."""
    n, k = len(keys), 2
    while k <= n:
        j = k // 2
        while j > 0:
            for i in range(n):
                l = i ^ j
                if l > i and l < n:
                    if ((i & k) == 0 and keys[i] > keys[l]) or ((i & k) != 0 and keys[i] < keys[l]):
                        keys[i], keys[l] = keys[l], keys[i]
            j //= 2
        k *= 2
    return keys


@pytest.mark.parametrize("mode", range(1, 8))
def test_sorting_modes_of_the_oracle_against_plain_python(mode):
    W, H, max_frags = 24, 16, 40
    c, P, nodes, start, lists = random_lists(W, H, max_frags, seed=100 + mode)
    P.ppllSortingMode = mode
    got = lvo.ppll_resolve(P, nodes, start, literal=False)
    bg = [float(b) for b in P.background[:]]
    unsorted_bitonic = 0
    for (x, y), (col, depth) in lists.items():
        keys = [(float(d), int(cc)) for d, cc in zip(depth, col)]
        if mode == 5:
            want_keys = bitonic_as_written(list(keys))
            unsorted_bitonic += want_keys != sorted(keys)
        else:
            want_keys = sorted(keys)
        assert got[y, x].tolist() == blend_all([k[1] for k in want_keys], bg), (mode, x, y, len(keys))
    if mode == 5:
        assert unsorted_bitonic > 0          # the reference's network does not sort lengths that are no power of two ...
        c2, P2, n2, s2, l2 = random_lists(8, 8, 64, seed=7, lengths=[1, 2, 4, 8, 16, 32, 64])
        P2.ppllSortingMode = 5               # ... and sorts the ones that are
        g2 = lvo.ppll_resolve(P2, n2, s2, literal=False)
        for (x, y), (col, depth) in l2.items():
            keys = sorted((float(d), int(cc)) for d, cc in zip(depth, col))
            assert g2[y, x].tolist() == blend_all([k[1] for k in keys], [float(b) for b in P2.background[:]])


def test_literal_depth_only_comparisons_sort_by_depth():
    """literal = the shaders' depth-only comparisons: with pairwise different depths every mode but the bitonic one gives
    the image of the key order."""
    W, H, max_frags = 16, 8, 33
    c, P, nodes, start, lists = random_lists(W, H, max_frags, seed=5)
    nodes = nodes.copy()
    nodes[:, 1] = (np.float32(0.2) + np.random.default_rng(1).permutation(len(nodes)).astype(np.float32) * np.float32(1e-4)).view(np.uint32)
    for mode in (1, 2, 3, 4, 6, 7):
        P.ppllSortingMode = mode
        assert np.array_equal(lvo.ppll_resolve(P, nodes, start, literal=True), lvo.ppll_resolve(P, nodes, start, literal=False)), mode


def test_the_priority_queue_differs_only_by_its_early_out():
    W, H, max_frags = 16, 8, 20
    c, P, nodes, start, lists = random_lists(W, H, max_frags, seed=9)
    faint = nodes.copy()
    faint[:, 0] &= np.uint32(0x1FFFFFFF)      # alpha <= 0.122 per layer, 20 layers: accumulated alpha stays below 0.99
    P.ppllSortingMode = 0
    pq = lvo.ppll_resolve(P, faint, start)
    P.ppllSortingMode = 2
    assert np.array_equal(lvo.ppll_resolve(P, faint, start), pq)
    opaque = nodes.copy()
    opaque[:, 0] |= np.uint32(0xFF000000)
    P.ppllSortingMode = 0
    a = lvo.ppll_resolve(P, opaque, start)
    P.ppllSortingMode = 4
    assert np.array_equal(lvo.ppll_resolve(P, opaque, start), a)   # opaque front layer: later layers weigh (1 - 1) = 0 anyway


@pytest.mark.gpu
@pytest.mark.parametrize("max_frags", [40, 150])       # fragment arrays in LDS / in the global scratch slab
def test_resolve_kernel_sorting_modes_on_given_lists(hip_lib, max_frags):
    c, P, nodes, start, lists = random_lists(40, 24, max_frags, seed=31 + max_frags)
    ctx = c.hip_context()
    for mode, name in enumerate(MODES):
        ctx.set_option("sorting_mode", name)
        P.ppllSortingMode = mode
        assert np.array_equal(ctx.ppll_resolve(nodes, start), lvo.ppll_resolve(P, nodes, start)), name
    ctx.set_option("sorting_mode", "5")                 # the index is accepted too
    P.ppllSortingMode = 5
    assert np.array_equal(ctx.ppll_resolve(nodes, start), lvo.ppll_resolve(P, nodes, start))
    with pytest.raises(Exception):
        ctx.set_option("sorting_mode", "Bogo Sort")


@pytest.mark.gpu
def test_whole_frames_in_every_sorting_mode(hip_lib):
    """Gather + resolve through the C-ABI: the oracle's frame of the same mode within the frame tolerance (the fragment multiset is
    bit-exact, test_gpu_parity; with (depth, colour) keys a complete sort makes the list order immaterial).  The bitonic network
    leaves lists whose length is no power of two partly unsorted, so ITS image depends on the order the fragments were linked in --
    the rasterisation order in the reference, the atomics' order here: that mode is checked on the lists the gather produced."""
    c = small_case(width=96, height=64, n_lines=50, line_width=0.03, transparent=True)
    sc = c.oracle_scene()
    ctx = c.hip_context()
    pw, ph = c.padded()
    frames = []
    for mode, name in enumerate(MODES):
        c.settings["sorting_mode"] = name
        ctx.set_option("sorting_mode", name)
        P = c.oracle_params(sc)
        assert P.ppllSortingMode == mode
        got = ctx.render(2)
        if mode == 5:
            nodes, start, _ = ctx.ppll_buffers(pw * ph, int(ctx.stats().ppll_pool_nodes))
            assert np.array_equal(got, lvo.ppll_resolve(P, nodes, start)), name
        else:
            want = sc.render_ppll(P)
            assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 2, name
        frames.append(got)
    for mode in (1, 2, 3, 4, 6, 7):           # the complete sorts agree with each other byte for byte
        assert np.array_equal(frames[mode], frames[1])
