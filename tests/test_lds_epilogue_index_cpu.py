"""Index arithmetic of the four-wave kernel's row-coalesced fp32 epilogue (csrc/conv_gemm.hip: conv_co_dump /
conv_epilogue_f32_lds), restated on the host: the dump (lane = output row, XOR-swizzled 512-B rows) and the transposed
read-back (lane L = row 8k + (L >> 3), columns 4 (L & 7) .. +3 of every 32-column block) must address the same bytes for the
same (row, column), cover the 64 x 128 half tile exactly once, be free of LDS bank conflicts in both directions, make every
wave-wide global access 8 whole 128-B lines, and the GroupNorm writer lanes must cover every group of the wave tile exactly
once for each statistics granularity.  (The kernel itself is held bit for bit to the 8-wave kernel on the GPU:
tests/test_kernels_gpu.py::test_w4_kernel_bit_identical_to_8wave_kernel; this file guards the arithmetic the comments state.)"""
import itertools

CO_ROW = 512


def dump_addr(l32, hi32, ni, mi, g):
    return (l32 + mi * 32) * CO_ROW + ni * 128 + (((2 * g + hi32) ^ (l32 & 7)) << 4)


def dump_elems(l32, hi32, ni, mi, g):
    row = mi * 32 + l32
    c0 = ni * 32 + 8 * g + 4 * hi32
    return [(row, c0 + j) for j in range(4)]


def read_addr(lane, ni, k):
    tr, tq = lane >> 3, lane & 7
    return tr * CO_ROW + ((tq ^ tr) << 4) + k * 8 * CO_ROW + ni * 128


def read_elems(lane, ni, k):
    tr, tq = lane >> 3, lane & 7
    return [(8 * k + tr, ni * 32 + 4 * tq + j) for j in range(4)]


def test_dump_and_read_back_address_the_same_bytes_once():
    where = {}
    for l32, hi32, ni, mi, g in itertools.product(range(32), range(2), range(4), range(2), range(4)):
        a = dump_addr(l32, hi32, ni, mi, g)
        assert 0 <= a < 64 * CO_ROW and a % 16 == 0
        for j, e in enumerate(dump_elems(l32, hi32, ni, mi, g)):
            assert e not in where
            where[e] = a + 4 * j
    assert len(where) == 64 * 128
    seen = set()
    for lane, ni, k in itertools.product(range(64), range(4), range(8)):
        a = read_addr(lane, ni, k)
        for j, e in enumerate(read_elems(lane, ni, k)):
            assert where[e] == a + 4 * j, (lane, ni, k, e)
            seen.add(e)
    assert len(seen) == 64 * 128


def _conflict_free(addrs):
    """A b128 LDS access is served 8 lanes at a time (8 x 16 B = the 128 B / clk of the LDS): the eight 16-B slots of each group of 8
    consecutive lanes must fall into eight different 16-B bank groups (address / 16 mod 8)."""
    for i in range(0, 64, 8):
        assert len({(a >> 4) & 7 for a in addrs[i:i + 8]}) == 8, addrs[i:i + 8]


def test_no_lds_bank_conflicts_either_way():
    for ni, mi, g in itertools.product(range(4), range(2), range(4)):
        _conflict_free([dump_addr(lane & 31, lane >> 5, ni, mi, g) for lane in range(64)])
    for ni, k in itertools.product(range(4), range(8)):
        _conflict_free([read_addr(lane, ni, k) for lane in range(64)])


def test_every_global_access_is_eight_whole_cache_lines():
    out_stride = 512                                     # fp32 elements per output row (any multiple of 32 keeps rows line-aligned)
    for ni, k in itertools.product(range(4), range(8)):
        lines = {}
        for lane in range(64):
            row, col = read_elems(lane, ni, k)[0]
            byte = (row * out_stride + col) * 4
            lines.setdefault(byte // 128, set()).update(range(byte % 128, byte % 128 + 16))
        assert len(lines) == 8 and all(len(v) == 128 for v in lines.values())
    # the lane-per-row form it replaces: 32 lines, a quarter of each
    lines = {}
    for l32, hi32 in itertools.product(range(32), range(2)):
        byte = (l32 * out_stride + 4 * hi32) * 4
        lines.setdefault(byte // 128, set()).update(range(byte % 128, byte % 128 + 16))
    assert len(lines) == 32 and all(len(v) == 32 for v in lines.values())


def test_groupnorm_writer_lanes_cover_every_group_once():
    nw0 = 128                                            # second wave tile of a row of tiles
    for cl in range(2, 8):                               # log2(channels per group): 4 .. 128
        groups = []
        for lane in range(64):
            tr, tq = lane >> 3, lane & 7
            if cl == 7:
                writer, grp = lane == 0, nw0 >> 7
            elif cl == 6:
                writer, grp = tr < 2 and tq == 0, (nw0 >> 6) + tr
            else:
                qpg = 1 << (cl - 2)
                writer, grp = tr < 4 and (tq & (qpg - 1)) == 0, (nw0 + tr * 32 + 4 * tq) >> cl
            if writer:
                groups.append(grp)
        first = nw0 >> cl
        assert sorted(groups) == list(range(first, first + (128 >> cl))), (cl, groups)
