"""CPU: the HOST backend of the transcript sponges (csrc/poseidon_sponge_host.hpp through h2agg_poseidon_squeeze_batch_host;
no device involved) against the oracle restatement of PoseidonChip (hash/poseidon.rs:144-231, T = 9, RATE = 8, R_F = 8,
R_P = 63), whose parameter generator reproduces the published poseidonperm_x5_254 vectors (tests/test_oracle_poseidon.py).
The same constants (poseidon_host::Spec) feed the device sponge; tests/test_gpu_poseidon.py and tests/test_gpu_verifier.py
check that the two backends agree bit for bit."""
import pytest

from oracle import bn254 as O
from oracle import poseidon as P


def fe(xs):
    return b"".join(O.fe_to_bytes(x) for x in xs)


def oracle_squeezes(row, upto):
    c, pos, out = P.PoseidonChip(), 0, []
    for u in upto:
        c.update(row[pos:u])
        pos = u
        out.append(c.squeeze())
    return out


@pytest.mark.parametrize("nelem", [0, 1, 7, 8, 9, 16, 17, 40])
def test_padding_and_chunking(pkg, nelem):
    rng = O.SplitMix64(0xA05 + nelem)
    nproofs = 5
    rows = [[rng.fr() for _ in range(nelem)] for _ in range(nproofs)]
    if nelem:
        rows[1][0] = 0
        rows[2][nelem - 1] = O.R - 1
    got = pkg.poseidon_squeeze_batch_host(b"".join(fe(r) for r in rows), nproofs, [nelem, nelem])
    for i, r in enumerate(rows):
        assert got[64 * i:64 * i + 64] == fe(oracle_squeezes(r, [nelem, nelem])), (nelem, i)


def test_interleaved_squeezes_and_threads(pkg):
    """absorb 3, squeeze, absorb 10, squeeze twice, absorb 8, squeeze; more proofs than worker threads; 1 thread == many"""
    rng = O.SplitMix64(0xA06)
    nproofs = 2 * pkg.host_threads() + 3
    rows = [[rng.fr() for _ in range(21)] for _ in range(nproofs)]
    upto = [3, 13, 13, 21]
    blob = b"".join(fe(r) for r in rows)
    got = pkg.poseidon_squeeze_batch_host(blob, nproofs, upto)
    assert got == pkg.poseidon_squeeze_batch_host(blob, nproofs, upto, max_threads=1)
    for i in (0, 1, nproofs // 2, nproofs - 1):
        assert got[128 * i:128 * i + 128] == fe(oracle_squeezes(rows[i], upto)), i


def test_scalar_and_ifma_kernels_agree(pkg):
    """the portable 4 x 64-bit permutation and the AVX-512 IFMA one (taken when the CPU has avx512ifma) are differential
    partners: same challenges on long streams, including the chunk edges"""
    rng = O.SplitMix64(0xA07)
    rows = [[rng.fr() for _ in range(203)] for _ in range(6)]
    rows[0][:8] = [0] * 8
    rows[1][:9] = [O.R - 1] * 9
    blob, upto = b"".join(fe(r) for r in rows), [0, 0, 1, 8, 9, 64, 64, 203]
    a = pkg.poseidon_squeeze_batch_host(blob, len(rows), upto, kernel="scalar")
    b = pkg.poseidon_squeeze_batch_host(blob, len(rows), upto, kernel="ifma")
    assert a == b
    assert a[:32 * len(upto)] == fe(oracle_squeezes(rows[0], upto))
    assert pkg.host_sponge_kind() in ("ifma", "scalar")


def test_extreme_values(pkg):
    """all-zero, all r-1 and 2^k-shaped elements: every conditional subtraction of the lazy dot products is exercised"""
    rows = [[0] * 24, [O.R - 1] * 24, [(1 << (11 * k % 253)) % O.R for k in range(24)], [O.R - 1 - k for k in range(24)]]
    got = pkg.poseidon_squeeze_batch_host(b"".join(fe(r) for r in rows), len(rows), [8, 24])
    for i, r in enumerate(rows):
        assert got[64 * i:64 * i + 64] == fe(oracle_squeezes(r, [8, 24])), i


def test_rejects_non_canonical_and_bad_positions(pkg):
    with pytest.raises(pkg.H2AggError) as ei:
        pkg.poseidon_squeeze_batch_host(O.R.to_bytes(32, "little"), 1, [1])
    assert ei.value.code == pkg.ERR_NONCANONICAL
    with pytest.raises(pkg.H2AggError):
        pkg.poseidon_squeeze_batch_host(bytes(64), 1, [2, 1])       # decreasing
    with pytest.raises(pkg.H2AggError):
        pkg.poseidon_squeeze_batch_host(bytes(64), 1, [3])          # beyond the stream


def test_worker_pool_serves_concurrent_callers(pkg):
    """several caller threads at once (contexts on different threads share the pool: one call's sponges run while another call
    is on the device): every caller gets its own batch's challenges, whatever the interleaving of the runs"""
    import threading
    rng = O.SplitMix64(0xA09)
    ncallers, nproofs = 6, 7
    batches = []
    for c in range(ncallers):
        rows = [[rng.fr() for _ in range(30 + c)] for _ in range(nproofs)]
        batches.append((b"".join(fe(r) for r in rows), [5, 30 + c]))
    want = [pkg.poseidon_squeeze_batch_host(blob, nproofs, upto, max_threads=1) for blob, upto in batches]
    got = [[None] * 8 for _ in range(ncallers)]

    def run(c):
        blob, upto = batches[c]
        for rep in range(8):
            got[c][rep] = pkg.poseidon_squeeze_batch_host(blob, nproofs, upto)

    ts = [threading.Thread(target=run, args=(c,)) for c in range(ncallers)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in range(ncallers):
        assert all(g == want[c] for g in got[c]), c
