"""CPU: the oracle's Poseidon against the PUBLISHED permutation vectors (poseidonperm_x5_254_3 / _5 — the vectors the
`poseidon` crate pinned by the reference asserts in its own unit tests), the optimized schedule of the reference's
`permutation` (hash/poseidon.rs:193-230) against the textbook permutation, and the sponge's padding rules
(hash/poseidon.rs:45-86,171-191)."""
import json
import os

from oracle import bn254 as O
from oracle import poseidon as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "poseidon_kats.json")) as f:
    KATS = json.load(f)


def test_published_permutation_vectors():
    assert P.spec(3, 8, 57).round_constants[0][0] == int(KATS["first_round_constant_x5_254_3"], 16)
    for k in KATS["perm"]:
        sp = P.spec(k["t"], k["r_f"], k["r_p"])
        assert P.permute_textbook(sp, list(range(k["t"]))) == [int(x, 16) for x in k["out"]]


def test_optimized_schedule_equals_textbook():
    rng = O.SplitMix64(5)
    for t, rf, rp in [(3, 8, 57), (5, 8, 60), (9, 8, 63)]:
        sp = P.spec(t, rf, rp)
        assert len(sp.start) == rf // 2 + 1 and len(sp.partial) == rp and len(sp.end) == rf // 2 - 1
        assert len(sp.sparse) == rp
        for ninp in (0, 1, t - 2, t - 1):
            st = [rng.fr() for _ in range(t)]
            inp = [rng.fr() for _ in range(ninp)]
            pre = list(st)
            for i, x in enumerate(inp):
                pre[i + 1] = (pre[i + 1] + x) % O.R
            if ninp + 1 < t:
                pre[ninp + 1] = (pre[ninp + 1] + 1) % O.R            # the padding one (poseidon.rs:72-82)
            assert P.permute_textbook(sp, pre) == P.permutation(sp, st, inp), (t, ninp)


def test_mds_is_invertible_cauchy():
    sp = P.spec(9, 8, 63)
    mi = P.mat_inv(sp.mds)
    ident = P.mat_mul(sp.mds, mi)
    assert all(ident[i][j] == (1 if i == j else 0) for i in range(9) for j in range(9))


def test_sponge_padding_rules():
    """squeeze (poseidon.rs:171-191): chunks of RATE; a full last chunk (or no input at all) costs one more
    permutation of the empty chunk; the state carries over between squeezes."""
    sp = P.spec(9, 8, 63)
    rng = O.SplitMix64(9)
    for n in (0, 1, 7, 8, 9, 16, 17):
        xs = [rng.fr() for _ in range(n)]
        c = P.PoseidonChip()
        c.update(xs)
        got = c.squeeze()
        st = [1 << 64] + [0] * 8
        chunks = [xs[i:i + 8] for i in range(0, n, 8)]
        for ch in chunks:
            st = P.permutation(sp, st, ch)
        if not chunks or len(chunks[-1]) == 8:
            st = P.permutation(sp, st, [])
        assert got == st[1]
        assert c.squeeze() == P.permutation(sp, st, [])[1]             # squeeze again: empty absorb, same state


def test_transcript_read_write_roundtrip():
    rng = O.SplitMix64(77)
    w = P.PoseidonTranscriptWrite()
    pts = [O.scalar_mul(rng.fr(), O.G1) for _ in range(5)] + [O.INF]
    scs = [rng.fr() for _ in range(4)]
    chal = []
    for p in pts[:3]:
        w.write_point(p)
    chal.append(w.squeeze_challenge_scalar())
    for s in scs:
        w.write_scalar(s)
    chal.append(w.squeeze_challenge_scalar())
    chal.append(w.squeeze_challenge_scalar())
    for p in pts[3:]:
        w.write_point(p)
    chal.append(w.squeeze_challenge_scalar())
    r = P.PoseidonTranscriptRead(w.finalize())
    got = [r.read_point() for _ in range(3)]
    c0 = r.squeeze_challenge_scalar()
    got_s = [r.read_scalar() for _ in range(4)]
    c1, c2 = r.squeeze_challenge_scalar(), r.squeeze_challenge_scalar()
    got += [r.read_point() for _ in range(3)]
    c3 = r.squeeze_challenge_scalar()
    assert got == pts and got_s == scs and [c0, c1, c2, c3] == chal
    try:
        r.read_point()
        assert False
    except P.TranscriptError:
        pass
