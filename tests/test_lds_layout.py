"""CPU checks of the fused kernel's data layout (no GPU): the LDS swizzle is a
bijection and makes every pass bank-conflict free under the gfx950 LDS rules
(MI355X_MICROARCH.md §LDS: ds_read_b64 = 2 groups of 32 lanes over 64 four-byte
banks; ds_write_b64 = 4 groups of 16 lanes over 32 banks), the thread -> block
pairing covers every block exactly once with mirror-image frequency sets, and
the thread-ordered spectrum tables are a permutation of the natural ones."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import GOLDEN



def extra_cycles(words, group, nbanks):
    """words: LDS word (8-byte) index per lane of one wave instruction. Returns the
    number of extra LDS cycles (0 = conflict free) summed over the lane groups."""
    extra = 0
    for g0 in range(0, len(words), group):
        banks = {}
        for a in set(words[g0:g0 + group]):
            for d in (2 * a, 2 * a + 1):
                banks.setdefault(d % nbanks, set()).add(d)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def pass_words(ups, m, tables):
    """Yield, per pass, the list of per-wave-instruction lane address lists."""
    K = 1 << m
    J, T = K // 16, K // 32
    rad = ups.fused_plan_radices(m)  # 2^(m mod 4), 16, .., 16
    assert np.prod(rad) == K and rad[-1] == 16
    block_b = tables["blockB"]
    L = K
    for pi, R in enumerate(rad):
        S = L // R
        last = pi == len(rad) - 1
        per_thread = max((K // R) // T, 1)
        instrs = []
        for w0 in range(0, T, 64):
            lanes = range(w0, min(w0 + 64, T))
            for bi in range(per_thread):
                for t in range(R):
                    words = []
                    for tau in lanes:
                        if last:
                            blk = ups.lib.mi_fused_block_a(tau, m) if bi == 0 else int(block_b[tau])
                            i = 16 * blk + t
                        else:
                            q = tau + bi * T
                            i = (q // S) * L + (q % S) + t * S
                        words.append(ups.lib.mi_lds_swizzle(i))
                    instrs.append(words)
        yield R, S, instrs
        L //= R


@pytest.mark.parametrize("m", [5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_swizzle_is_a_bijection_on_every_size(ups, m):
    K = 1 << m
    assert sorted(ups.lib.mi_lds_swizzle(i) for i in range(K)) == list(range(K))


@pytest.mark.parametrize("m,fname", [(12, "filter_48k_16x_80000_min_phase.json"), (14, "filter_44k_4x_80000_min_phase.json")])
def test_every_pass_is_bank_conflict_free(ups, m, fname):
    tables = ups.build_tables(GOLDEN / "filters" / fname)
    assert tables["geometry"]["log2k"] == m
    for R, S, instrs in pass_words(ups, m, tables):
        full = [w for w in instrs if len(w) == 64]
        assert full, (R, S)
        rd = sum(extra_cycles(w, 32, 64) for w in full)
        wr = sum(extra_cycles(w, 16, 32) for w in full)
        assert rd == 0 and wr == 0, f"radix {R} stride {S}: {rd} read / {wr} write conflict cycles"


@pytest.mark.parametrize("m", [5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_block_pairing_is_a_mirror_partition(ups, make_filter, m):
    K = 1 << m
    J, T = K // 16, K // 32
    # a filter whose geometry gives K complex points: L = 1, fft = 2K
    fft = 2 * K
    taps = fft // 4 + 1
    t = ups.build_tables(make_filter(np.ones(taps, np.float32), fft, fft - (taps - 1), 1, name=f"m{m}"))
    assert t["geometry"]["log2k"] == m and t["blockB"].size == T
    sets = [ups.lib.mi_fused_set_of_block(b, m) for b in range(J)]
    assert sorted(sets) == list(range(J))  # digit reversal is a permutation
    seen = []
    for tau in range(T):
        a_blk, b_blk = ups.lib.mi_fused_block_a(tau, m), int(t["blockB"][tau])
        a, b = sets[a_blk], sets[b_blk]
        if tau == 0:
            assert (a, b) == (0, J // 2)
        else:
            assert 0 < a < J // 2 and b == J - a  # S_a and its mirror S_{J-a}
        seen += [a_blk, b_blk]
    assert sorted(seen) == list(range(J))  # every block owned by exactly one thread


def test_thread_ordered_tables_are_a_permutation_of_the_natural_ones(ups):
    t = ups.build_tables(GOLDEN / "filters" / "filter_44k_4x_80000_min_phase.json")
    g = t["geometry"]
    K, P, m = g["K"], g["P"], g["log2k"]
    J, T = K // 16, K // 32
    Gs, Gc = t["Gs"].reshape(P, K), t["Gc"].reshape(P, K)
    GT = t["GT"].reshape(P, 16, T, 2)
    G0 = t["G0"].reshape(P, 17, 2)
    for tau in (1, 2, 17, 255, T - 1):
        a = ups.lib.mi_fused_set_of_block(ups.lib.mi_fused_block_a(tau, m), m)
        assert t["WmT"][tau] == t["Wm"][a]
        for p in range(P):
            for s in range(16):
                k = a + s * J
                assert GT[p, s, tau, 0] == Gs[p, k] and GT[p, s, tau, 1] == Gc[p, k]
    # thread 0: the self-mirrored sets S_0 (pairs 0..8) and S_{J/2} (pairs 0..7)
    for p in range(P):
        ks = [s * J for s in range(9)] + [J // 2 + s * J for s in range(8)]
        for s, k in enumerate(ks):
            assert G0[p, s, 0] == Gs[p, k] and G0[p, s, 1] == Gc[p, k]
