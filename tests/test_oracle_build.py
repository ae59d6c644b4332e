"""CPU: the oracle's build half against the golden fixtures, the figures measured from the
reference's own host code, and structural properties of the tree."""
import numpy as np
import pytest

from oracle import orc
from conftest import SCENES, make_soup

# Figures the reference host code (scene.cpp + ray_tracer.cpp::loadScene) produced for its
# four scenes: triangle count, shapes, materials incl. the appended default, distinct Morton
# codes, largest code.  They pin flattening, scene box and Morton arithmetic.
REFERENCE_FACTS = {
    "scene01": dict(n=862, shapes=4, materials=7, distinct=257, max_code=18204,
                    bb=([-5.764, 0, -5], [5.764, 10, 5])),
    "scene02": dict(n=1732, shapes=10, materials=10, distinct=50, max_code=3343,
                    bb=([-14.41, -0.225, 0], [14.41, 15.15, 21.74])),
    "scene03": dict(n=1732, shapes=10, materials=7, distinct=50, max_code=3343,
                    bb=([-12.55, -2.005, -7.581], [16.28, 13.37, 14.16])),
    "scene04": dict(n=2416, shapes=10, materials=9, distinct=318, max_code=25162,
                    bb=([-10.28, 0, -10.4], [10.2, 15.9, 9.9])),
}
# right spine of the tree down to the out-of-range leaf 2n
SPINES = {862: [0, 512, 768, 832, 848, 856, 860, 1724],
          1732: [0, 1024, 1536, 1664, 1728, 3464],
          2416: [0, 2048, 2304, 2368, 2400, 4832]}


@pytest.mark.parametrize("scene", SCENES)
def test_reference_facts(golden, scene):
    g, f = golden[scene], REFERENCE_FACTS[scene]
    assert len(g["prims"]) == f["n"]
    assert g["parse_counts"][2] == f["shapes"] and g["parse_counts"][3] == f["materials"]
    b = orc.build(g["prims"])
    codes = b["rays"]["morton_code"]
    assert len(np.unique(codes)) == f["distinct"]
    assert int(codes.max()) == f["max_code"] and int(codes.max()) < 2 ** 15
    np.testing.assert_allclose(b["bb"]["min"][0], f["bb"][0], atol=6e-3)
    np.testing.assert_allclose(b["bb"]["max"][0], f["bb"][1], atol=6e-3)
    assert np.all(g["prims"]["pad"] == -1)


@pytest.mark.parametrize("scene", SCENES)
def test_build_matches_golden(golden, scene):
    g = golden[scene]
    b = orc.build(g["prims"])
    assert np.array_equal(b["rays"]["morton_code"], g["morton"])
    assert np.array_equal(b["sorted_rays"]["index"], g["perm"])
    assert b["tree"].tobytes() == g["tree"].tobytes()
    assert b["bb"].tobytes() == g["bb"].tobytes()
    # sorted order: ascending codes, ties by ascending primitive index
    k, p = b["sorted_rays"]["morton_code"], b["sorted_rays"]["index"]
    assert np.all((k[:-1] < k[1:]) | ((k[:-1] == k[1:]) & (p[:-1] < p[1:])))
    assert np.array_equal(np.sort(p), np.arange(len(p)))


def _check_tree_structure(n, tree):
    """n internal nodes, n real leaves + the out-of-range leaf 2n, each reachable exactly once."""
    left, right, parent = tree["left"], tree["right"], tree["parent"]
    seen = np.zeros(2 * n + 1, dtype=np.int32)
    stack = [0]
    while stack:
        i = stack.pop()
        seen[i] += 1
        if i < n:
            stack.append(int(right[i]))
            stack.append(int(left[i]))
            for c in (left[i], right[i]):
                if c < 2 * n:
                    assert parent[c] == i
    assert np.all(seen == 1)
    assert parent[0] == 0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 862, 1023, 1024, 1025, 1732, 2416, 4097, 16383, 16384, 100000])
def test_radix_tree_literal_equals_textbook(n):
    a = orc.radix_tree(n, literal=False)
    b = orc.radix_tree(n, literal=True)
    assert a.tobytes() == b.tobytes()
    _check_tree_structure(n, a)
    # right spine ends in the out-of-range leaf
    i, spine = 0, [0]
    while i < n:
        i = int(a["right"][i])
        spine.append(i)
    assert spine[-1] == 2 * n
    if n in SPINES:
        assert spine == SPINES[n]
    # depth of the deepest internal node = bit_length(n)
    depth = np.zeros(2 * n + 1, dtype=np.int32)
    order = [0]
    for i in order:
        if i < n:
            for c in (int(a["left"][i]), int(a["right"][i])):
                depth[c] = depth[i] + 1
                order.append(c)
    assert depth[:n].max() + 1 == n.bit_length()


def closed_form_nodes(n):
    """The tree over keys 0..n is the binary trie of the key indices: level k (split on bit k-1),
    block b covers keys [b 2^k, min((b+1) 2^k - 1, n)] and is a node iff b 2^k + 2^(k-1) <= n; its
    id is the last key when it is a left child (b even and (b+1) 2^k <= n), else the first key."""
    out = {}
    for k in range(1, n.bit_length() + 1):
        half = 1 << (k - 1)
        b = 0
        while (b << k) + half <= n:
            s = b << k
            idx = ((b + 1) << k) - 1 if (b % 2 == 0 and ((b + 1) << k) <= n) else s
            gamma = s + half - 1
            e = min(s + (1 << k) - 1, n)
            left = n + gamma if k == 1 else gamma
            right = n + gamma + 1 if gamma + 1 == e else gamma + 1
            assert idx not in out
            out[idx] = (left, right, k)
            b += 1
    return out


@pytest.mark.parametrize("n", list(range(1, 70)) + [127, 128, 129, 862, 1023, 1024, 1025, 2047, 2048, 2049, 2416, 5000])
def test_closed_form_enumeration_equals_radix_tree(n):
    tree = orc.radix_tree(n)
    nodes = closed_form_nodes(n)
    assert sorted(nodes) == list(range(n))
    for i, (l, r, _) in nodes.items():
        assert tree["left"][i] == l and tree["right"][i] == r
    # children sit on strictly lower levels (bottom-up level order is a valid refit schedule)
    for i, (l, r, k) in nodes.items():
        for c in (l, r):
            if c < n:
                assert nodes[c][2] < k


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 4097])
def test_refit_fixpoint(n):
    prims = make_soup(n, seed=n)
    b = orc.build(prims)
    t = b["tree"]
    for i in range(n):
        l, r = int(t["left"][i]), int(t["right"][i])
        assert np.array_equal(t["bound"]["min"][i], np.minimum(t["bound"]["min"][l], t["bound"]["min"][r]))
        assert np.array_equal(t["bound"]["max"][i], np.maximum(t["bound"]["max"][l], t["bound"]["max"][r]))
    # leaves hold the sorted primitives; the out-of-range leaf is the zero box
    assert np.array_equal(t["index"][n:2 * n], b["sorted_cs"]["index"])
    assert t[2 * n].tobytes() == bytes(48)
    # the root therefore contains the origin and the scene box
    assert np.all(t["bound"]["min"][0] <= 0) and np.all(t["bound"]["max"][0] >= 0)
    assert np.all(t["bound"]["min"][0] <= b["bb"]["min"][0]) and np.all(t["bound"]["max"][0] >= b["bb"]["max"][0])


def test_morton_bit_layout():
    """x gets the highest bit of each triple; quantisation is to 22 levels (0..21) per axis."""
    big = np.zeros(1, dtype=orc.BOUND)
    big["min"][0] = (0, 0, 0)
    big["max"][0] = (1, 1, 1)
    small = np.zeros(1, dtype=orc.BOUND)
    import ctypes as C
    def code(p):
        small["min"][0] = p
        small["max"][0] = p
        return orc.lib().orc_morton(small.ctypes.data_as(C.c_void_p), big.ctypes.data_as(C.c_void_p))
    assert code((0, 0, 0)) == 0
    assert code((1 / 21 + 1e-4, 0, 0)) == 4          # a = 1 -> bit 2
    assert code((0, 1 / 21 + 1e-4, 0)) == 2          # b = 1 -> bit 1
    assert code((0, 0, 1 / 21 + 1e-4)) == 1          # c = 1 -> bit 0
    assert code((1, 1, 1)) == 0b111_000_111_000_111  # 21 = 0b10101 on every axis


def test_morton_flat_axis_is_cell_zero():
    """A scene box that is flat on an axis gives 0/0 there; the x86 conversion the reference relies on
    leaves zero in the 22 bits that are interleaved (structs.hpp:81-105), and so does the oracle."""
    import ctypes as C
    big = np.zeros(1, dtype=orc.BOUND)
    big["min"][0] = (0, 0.5, 0)
    big["max"][0] = (1, 0.5, 1)                       # flat in y
    small = np.zeros(1, dtype=orc.BOUND)
    small["min"][0] = (1, 0.5, 1)
    small["max"][0] = (1, 0.5, 1)
    code = orc.lib().orc_morton(small.ctypes.data_as(C.c_void_p), big.ctypes.data_as(C.c_void_p))
    assert code == 0b101_000_101_000_101              # 21 on x and z, 0 on y


def _split_tree(keys, n):
    """Binary radix tree of n sorted distinct keys by recursive splitting at the highest differing bit (the
    definition; Karras' searches are a way to find the same splits in parallel): node = first key of the range for a
    right child, last key for a left child, root 0; leaves n + key."""
    left, right = {}, {}

    def build(lo, hi, node):
        if lo == hi:
            return n + lo
        top = (keys[lo] ^ keys[hi]).bit_length() - 1                     # highest bit in which the range differs
        split = lo
        while ((keys[split + 1] >> top) & 1) == ((keys[lo] >> top) & 1):  # last key that shares the first key's bit
            split += 1
        left[node] = build(lo, split, split)
        right[node] = build(split + 1, hi, split + 1)
        return node
    build(0, n - 1, 0)
    return left, right


@pytest.mark.parametrize("n,distinct", [(2, 2), (3, 1), (17, 5), (64, 64), (500, 37), (2000, 2000), (2000, 1)])
def test_radix_tree_over_codes_equals_recursive_splitting(n, distinct):
    """NON-PARITY option PT_BUILD_TOPOLOGY_MORTON: the oracle's search-based tree over sorted codes (equal codes told
    apart by position) against the tree's definition, with codes that use all 63 bits and with heavy ties."""
    import ctypes as C
    rng = np.random.default_rng(n * 1000 + distinct)
    pool = rng.integers(0, 2 ** 63 - 1, size=distinct, dtype=np.uint64)
    codes = np.sort(rng.choice(pool, size=n))
    tree = np.zeros(2 * n + 1, dtype=orc.TREENODE)
    orc.lib().orc_radix_tree_codes(C.c_int(n), codes.ctypes.data_as(C.c_void_p), tree.ctypes.data_as(C.c_void_p))
    keys = [(int(c) << 32) | i for i, c in enumerate(codes)]             # the (code, position) keys the searches compare
    left, right = _split_tree(keys, n)
    assert sorted(left) == list(range(n - 1))
    for i in range(n - 1):
        assert tree["left"][i] == left[i] and tree["right"][i] == right[i], i
        for c in (left[i], right[i]):
            assert tree["parent"][c] == i
    assert not tree[n - 1].tobytes().strip(b"\0") and not tree[2 * n].tobytes().strip(b"\0")
