"""Synthetic read/reference batches from a portable PRNG.

The generator is counter-based splitmix64, so a batch depends only on (seed, shape,
rates) -- not on libc, numpy's RNG version or the machine -- and fixtures regenerate
bit-identically anywhere.  Shapes follow the reference host's contract: every read is
exactly R bytes and every ref exactly F bytes, short sequences right-padded with '\\0'
(pad(), src/util/versalignUtil.cpp:17-33).
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(counter):
    """Hash uint64 counters -> uint64 (the splitmix64 output function)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(counter, dtype=np.uint64) + _GOLDEN)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, stream, shape):
    """Independent uint64 field for (seed, stream), one value per element of shape."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) * np.uint64(0x10001) + np.uint64(stream) * np.uint64(0xA5A5A5A5))
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D)
    return splitmix64(ctr).reshape(shape)


def _uniform(seed, stream, shape):
    return (_stream(seed, stream, shape) >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def make_pairs(n, R, F, seed=1, sub_rate=0.15, indel_rate=0.0, n_run_frac=0.01,
               short_frac=0.01, lowercase_frac=0.0, junk_frac=0.0):
    """-> reads uint8 [n,R], refs uint8 [n,F].

    ref: i.i.d. uniform ACGT.  read: ref[off:off+R] (off uniform in [0, F-R], or a
    prefix copy when R > F with the tail drawn at random), each base replaced by a
    uniform random base with probability sub_rate; with indel_rate > 0 single-base
    insertions/deletions are applied (per-pair loop: small n only).  n_run_frac of the
    pairs get a run of 1-5 'N' in the read, short_frac get a read AND a ref truncated
    and padded with NUL, lowercase_frac are lower-cased, junk_frac get one arbitrary
    byte (including values >= 0x80) somewhere in read and ref.
    """
    n, R, F = int(n), int(R), int(F)
    refs = BASES[(_stream(seed, 1, (n, F)) >> np.uint64(33)).astype(np.int64) & 3] if n * F else \
        np.zeros((n, F), dtype=np.uint8)
    reads = np.zeros((n, R), dtype=np.uint8)
    if n == 0 or R == 0:
        return reads, refs.copy()
    span = max(F - R, 0) + 1
    off = ((_stream(seed, 2, (n,)) >> np.uint64(20)) % np.uint64(span)).astype(np.int64)
    cols = np.arange(R, dtype=np.int64)[None, :] + off[:, None]
    rnd_base = BASES[(_stream(seed, 3, (n, R)) >> np.uint64(33)).astype(np.int64) & 3]
    inside = cols < F
    src = np.take_along_axis(refs, np.minimum(cols, max(F - 1, 0)), axis=1) if F else rnd_base
    reads = np.where(inside, src, rnd_base).astype(np.uint8)
    sub = _uniform(seed, 4, (n, R)) < sub_rate
    reads = np.where(sub, rnd_base, reads).astype(np.uint8)

    if indel_rate > 0:
        u = _uniform(seed, 5, (n, R))
        kind = (_stream(seed, 6, (n, R)) >> np.uint64(40)).astype(np.int64) & 1
        ins = BASES[(_stream(seed, 7, (n, R)) >> np.uint64(35)).astype(np.int64) & 3]
        for i in np.nonzero((u < indel_rate).any(axis=1))[0]:
            out = []
            for j in range(R):
                if u[i, j] < indel_rate:
                    if kind[i, j]:
                        out.append(ins[i, j])
                        out.append(reads[i, j])
                    # else: deletion, emit nothing
                else:
                    out.append(reads[i, j])
            out = out[:R] + [0] * max(0, R - len(out))
            reads[i] = np.asarray(out[:R], dtype=np.uint8)

    pick = _uniform(seed, 8, (n,))
    aux = _stream(seed, 9, (n, 4))
    for i in np.nonzero(pick < n_run_frac)[0]:
        run = 1 + int(aux[i, 0] % np.uint64(5))
        start = int(aux[i, 1] % np.uint64(max(R - run, 0) + 1))
        reads[i, start:start + run] = ord("N")
    for i in np.nonzero((pick >= n_run_frac) & (pick < n_run_frac + short_frac))[0]:
        keep_r = int(aux[i, 0] % np.uint64(R + 1))
        keep_f = int(aux[i, 1] % np.uint64(F + 1))
        reads[i, keep_r:] = 0
        refs[i, keep_f:] = 0
    lo = n_run_frac + short_frac
    for i in np.nonzero((pick >= lo) & (pick < lo + lowercase_frac))[0]:
        reads[i] = np.where(reads[i] >= 65, reads[i] | 0x20, reads[i])
        if aux[i, 2] & np.uint64(1):
            refs[i] = np.where(refs[i] >= 65, refs[i] | 0x20, refs[i])
    lo += lowercase_frac
    for i in np.nonzero((pick >= lo) & (pick < lo + junk_frac))[0]:
        reads[i, int(aux[i, 0] % np.uint64(R))] = np.uint8(aux[i, 2] & np.uint64(0xFF))
        if F:
            refs[i, int(aux[i, 1] % np.uint64(F))] = np.uint8(aux[i, 3] & np.uint64(0xFF))
    return np.ascontiguousarray(reads), np.ascontiguousarray(refs)


def make_ragged_pairs(n, R, F, seed=1, min_read_frac=0.1, min_ref_frac=0.1, trailing_n_frac=0.05, **kw):
    """make_pairs, then every read and ref keeps a uniformly drawn prefix and the rest becomes the
    NUL padding the reference host's pad() appends (src/util/versalignUtil.cpp:17-33): what a FASTA
    of mixed lengths looks like at the plugin boundary.  trailing_n_frac of the pairs end in a short
    run of 'N' before the padding."""
    reads, refs = make_pairs(n, R, F, seed=seed, **kw)
    if n == 0:
        return reads, refs
    u = _uniform(seed, 20, (n, 3))
    keep_r = np.minimum(R, (R * (min_read_frac + (1 - min_read_frac) * u[:, 0])).astype(np.int64) + 1)
    keep_f = np.minimum(F, (F * (min_ref_frac + (1 - min_ref_frac) * u[:, 1])).astype(np.int64) + 1)
    reads[np.arange(R)[None, :] >= keep_r[:, None]] = 0
    refs[np.arange(F)[None, :] >= keep_f[:, None]] = 0
    tail = u[:, 2] < trailing_n_frac
    for i in np.nonzero(tail)[0]:
        reads[i, max(0, keep_r[i] - 3):keep_r[i]] = ord("N")
        refs[i, max(0, keep_f[i] - 2):keep_f[i]] = ord("n")
    return np.ascontiguousarray(reads), np.ascontiguousarray(refs)


def cells(n, R, F):
    """DP cell updates of a batch: every pair costs exactly R*F (padding is computed)."""
    return int(n) * int(R) * int(F)


def gcups(n, R, F, seconds):
    return cells(n, R, F) / seconds / 1e9
