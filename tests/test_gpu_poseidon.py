"""GPU parity of the transcript side: the device Poseidon sponge (hash/poseidon.rs:144-231, T = 9,
RATE = 8, R_F = 8, R_P = 63), PoseidonEncode (mock/transcript_encode.rs:28-74) and the batched PoseidonTranscriptRead
(systems/halo2/transcript.rs:10-179) against the oracle restatement, whose parameter generator reproduces the published
poseidonperm_x5_254 vectors (tests/test_oracle_poseidon.py)."""
import pytest

from oracle import bn254 as O
from oracle import poseidon as P

pytestmark = pytest.mark.gpu


def fe(xs):
    return b"".join(O.fe_to_bytes(x) for x in xs)


@pytest.mark.parametrize("nelem", [0, 1, 7, 8, 9, 16, 17, 40])
def test_sponge_padding_and_chunking(eng, nelem):
    rng = O.SplitMix64(0x905 + nelem)
    nproofs = 5
    rows = [[rng.fr() for _ in range(nelem)] for _ in range(nproofs)]
    if nelem:
        rows[1][0] = 0
        rows[2][nelem - 1] = O.R - 1
    got = eng.poseidon_squeeze_batch(b"".join(fe(r) for r in rows), nproofs, [nelem, nelem]) if nelem else \
        eng.poseidon_squeeze_batch(b"", nproofs, [0, 0])
    for i, r in enumerate(rows):
        c = P.PoseidonChip()
        c.update(r)
        want = [c.squeeze(), c.squeeze()]
        assert got[64 * i:64 * i + 64] == fe(want), (nelem, i)


def test_sponge_interleaved_squeezes(eng):
    """absorb 3, squeeze, absorb 10, squeeze twice, absorb 8, squeeze — the state carries over (poseidon.rs:171-191)"""
    rng = O.SplitMix64(0x906)
    nproofs = 7
    rows = [[rng.fr() for _ in range(21)] for _ in range(nproofs)]
    got = eng.poseidon_squeeze_batch(b"".join(fe(r) for r in rows), nproofs, [3, 13, 13, 21])
    for i, r in enumerate(rows):
        c = P.PoseidonChip()
        want = []
        c.update(r[:3])
        want.append(c.squeeze())
        c.update(r[3:13])
        want.append(c.squeeze())
        want.append(c.squeeze())
        c.update(r[13:])
        want.append(c.squeeze())
        assert got[128 * i:128 * i + 128] == fe(want), i


def test_backends_agree(eng, pkg):
    """device kernel == host worker threads, through the same entry point (h2agg_transcript_configure), and the automatic
    choice returns the same bytes; the transcript reader likewise"""
    rng = O.SplitMix64(0x90B)
    nproofs = 9
    rows = [[rng.fr() for _ in range(37)] for _ in range(nproofs)]
    blob, upto = b"".join(fe(r) for r in rows), [0, 5, 16, 16, 37]
    got = {}
    try:
        for be in ("device", "host", "auto"):
            eng.transcript_configure(be)
            got[be] = eng.poseidon_squeeze_batch(blob, nproofs, upto)
    finally:
        eng.transcript_configure("auto")
    assert got["device"] == got["host"] == got["auto"] == pkg.poseidon_squeeze_batch_host(blob, nproofs, upto)


def test_sponge_rejects_non_canonical_elements(eng, pkg):
    bad = O.R.to_bytes(32, "little")
    with pytest.raises(pkg.H2AggError) as ei:
        eng.poseidon_squeeze_batch(bad, 1, [1])
    assert ei.value.code == pkg.ERR_NONCANONICAL


def test_transcript_read_batch_vs_oracle(eng, pkg):
    """the halo2 reading order of a small circuit: vk scalar, instance commitment, advice points, theta, ..., evals, v, W, u"""
    rng = O.SplitMix64(0x907)
    script = "CX" + "PPP" + "Q" + "PP" + "QQ" + "PPPP" + "Q" + "PP" + "Q" + "SSSSSSSSSSS" + "Q" + "PPP" + "Q" + "Q"
    nproofs = 6
    vk_scalar = rng.fr()
    proofs, ext, want_pts, want_ch = [], [], [], []
    for i in range(nproofs):
        w = P.PoseidonTranscriptWrite()
        inst = O.scalar_mul(rng.fr(), O.G1) if i != 2 else O.INF        # an identity instance commitment (empty column)
        pts, ch = [], []
        for op in script:
            if op == "C":
                w.common_scalar(vk_scalar)
            elif op == "X":
                w.common_point(inst)
            elif op == "P":
                p = O.scalar_mul(rng.fr(), O.G1) if (i, len(pts)) != (1, 4) else O.INF   # an identity in a proof
                w.write_point(p)
                pts.append(p)
            elif op == "S":
                w.write_scalar(rng.fr())
            else:
                ch.append(w.squeeze_challenge_scalar())
        proofs.append(w.finalize())
        ext.append(O.aff_to_bytes(inst))
        want_pts.append(b"".join(O.aff_to_bytes(p) for p in pts))
        want_ch.append(fe(ch))
        # the oracle reader agrees with the oracle writer
        r = P.PoseidonTranscriptRead(proofs[-1])
        r.common_scalar(vk_scalar)
        r.common_point(inst)
        rc = []
        for op in script[2:]:
            if op == "P":
                r.read_point()
            elif op == "S":
                r.read_scalar()
            else:
                rc.append(r.squeeze_challenge_scalar())
        assert rc == ch
    got_pts, got_ch = eng.transcript_read_batch(proofs, script, O.fe_to_bytes(vk_scalar), b"".join(ext))
    assert got_pts == want_pts
    assert got_ch == want_ch
    # "invalid point encoding in proof" / "invalid field element encoding in proof"
    bad = bytearray(proofs[0])
    bad[0:32] = (O.P - 1).to_bytes(32, "little")              # x = p - 1: x^3 + 3 = 2 is not a square mod p? decided below
    x = O.P - 1
    if pow((x * x * x + 3) % O.P, (O.P - 1) // 2, O.P) == 1:   # it IS a square: take another x that is not
        x = next(v for v in range(2, 50) if pow((v ** 3 + 3) % O.P, (O.P - 1) // 2, O.P) != 1)
        bad[0:32] = x.to_bytes(32, "little")
    with pytest.raises(pkg.BadPoint):
        eng.transcript_read_batch([bytes(bad)] + proofs[1:], script, O.fe_to_bytes(vk_scalar), b"".join(ext))
    first_scalar = 32 * script[:script.index("S")].count("P")
    bad = bytearray(proofs[0])
    bad[first_scalar:first_scalar + 32] = O.R.to_bytes(32, "little")
    with pytest.raises(pkg.H2AggError) as ei:
        eng.transcript_read_batch([bytes(bad)] + proofs[1:], script, O.fe_to_bytes(vk_scalar), b"".join(ext))
    assert ei.value.code == pkg.ERR_NONCANONICAL
    with pytest.raises(ValueError):
        eng.transcript_read_batch([proofs[0][:-32]], script, O.fe_to_bytes(vk_scalar), ext[0])   # short proof: read_exact
