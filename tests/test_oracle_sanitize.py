"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle sanitize): the golden vectors that pin it
(tests/test_oracle_c.py) and a 4 096-element random batch run through the sanitized build in a child process.  The oracle is the bulk
checker of every full-batch GPU test and of bench.py's parity gate; undefined behaviour in its unsigned __int128 arithmetic would be a
parity hole.  No GPU, no product code: test infrastructure checking test infrastructure."""
import os
import random
import subprocess

import numpy as np

import curve4q_oracle as o
import oracle_c as oc
from fourq_amd import codec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = 0x4F51524F55514652


def _record(op, kind, n, has_table, *arrays):
    """One record of the driver's file format (oracle/sanitize_main.c) as a list holding ONE flat array."""
    head = np.array([op, kind, n, has_table], dtype=np.uint64)
    return [np.concatenate([head] + [np.ascontiguousarray(a, dtype=np.uint64).ravel() for a in arrays])]


def _status_words(st):
    raw = np.zeros((len(st) + 7) // 8 * 8, dtype=np.uint8)
    raw[:len(st)] = st
    return raw.view(np.uint64)


def test_sanitized_oracle_on_golden_vectors_and_a_random_batch(golden, tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True, capture_output=True)
    exe = os.path.join(ROOT, "oracle", "_build", "fourq_oracle_sanitize")
    recs = []
    g = golden("mul.json")
    rows = g["var"] + g["edge"]
    s = codec.pack_scalars([r[0] for r in rows])
    p = codec.pack_points([r[1] for r in rows], 5)
    recs += _record(0, oc.ENDO, len(rows), 0, s, p, codec.pack_points([r[2] for r in rows], 5))
    recs += _record(0, oc.WINDOWED, len(rows), 0, s, p, codec.pack_points([r[3] for r in rows], 5))
    for blk in g["fixed"]:
        s = codec.pack_scalars([r[0] for r in blk["rows"]])
        recs += _record(0, oc.ENDO, len(blk["rows"]), 1, s, codec.pack_table(blk["table_endo"]), codec.pack_points([r[1] for r in blk["rows"]], 5))
        recs += _record(0, oc.WINDOWED, len(blk["rows"]), 1, s, codec.pack_table(blk["table_windowed"]), codec.pack_points([r[2] for r in blk["rows"]], 5))
    for Pt, tw, te in golden("tables.json")["tables"]:
        recs += _record(3, oc.WINDOWED, 1, 0, codec.pack_point(Pt), codec.pack_table(tw))
        recs += _record(3, oc.ENDO, 1, 0, codec.pack_point(Pt), codec.pack_table(te))
    # DH: golden rows, both rejections, and the edge scalars on the generator
    from conftest import unhex
    raw_all = golden("dh.json", raw=True)
    dh_rows = unhex(raw_all["dh"])
    s = codec.pack_scalars([r[0] for r in dh_rows])
    p = codec.pack_points([r[1] for r in dh_rows], 2)
    for kind, col in ((oc.ENDO, 2), (oc.WINDOWED, 3)):
        recs += _record(1, kind, len(dh_rows), 0, s, p, codec.pack_points([r[col] for r in dh_rows], 2), _status_words(np.zeros(len(dh_rows), np.uint8)))
    for m, Pt, msg in raw_all["reject"]:
        st = np.array([1 if msg == "Point not on curve" else 2], dtype=np.uint8)
        recs += _record(1, oc.ENDO, 1, 0, codec.pack_scalars([int(m, 16)]), codec.pack_points([unhex(Pt)], 2), np.zeros(8, np.uint64), _status_words(st))
    dec = golden("recode.json", raw=True)["decompose"]          # raw: the file also holds plain (negative) ints, which unhex() does not take
    recs += _record(2, 0, len(dec), 0, codec.pack_scalars([int(r[0], 16) for r in dec]), np.array([[int(w, 16) for w in r[1]] for r in dec], dtype=np.uint64))
    # a 4 096-element random variable-base batch (projective points: outputs of a fixed-base batch), expected words from the
    # regular build of the same source -- what the sanitizer adds is the absence of reports on inputs nobody hand-picked
    rng = random.Random(40960)
    n = 4096
    ks = codec.pack_scalars([rng.getrandbits(256) for _ in range(n)])
    ms = codec.pack_scalars([rng.getrandbits(256) for _ in range(n)])
    g1 = codec.pack_point(o.AffineToR1(o.Gx, o.Gy))
    pts = oc.mul(oc.ENDO, ks, None, oc.table(oc.ENDO, g1))
    recs += _record(0, oc.ENDO, n, 0, ms, pts, oc.mul(oc.ENDO, ms, pts))
    recs += _record(0, oc.WINDOWED, n, 0, ms, pts, oc.mul(oc.WINDOWED, ms, pts))
    path = tmp_path / "vectors.bin"
    with open(path, "wb") as fh:
        np.array([MAGIC, len(recs)], dtype=np.uint64).tofile(fh)
        np.concatenate(recs).tofile(fh)
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    proc = subprocess.run([exe, str(path)], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr[-4000:]
    assert "runtime error" not in proc.stderr and "AddressSanitizer" not in proc.stderr, proc.stderr[-4000:]
    assert "%d records" % len(recs) in proc.stdout and " 0 differ" in proc.stdout, proc.stdout
