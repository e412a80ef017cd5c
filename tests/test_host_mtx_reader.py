"""CPU: the C++ host's fast Matrix Market path (iterative_solvers_amd/host/mtx_fast.hpp: parallel parser,
counting sort, optional binary cache) must yield the SAME row-sorted COO as the fscanf reader that restates
the reference's read_mtx (io_funcs.hpp:125-289, mmio.h:195-226).  `its_solve --parse-only` prints
rows cols nnz and an order-sensitive checksum of (I, J, values); no GPU is touched."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
EXE = os.path.join(os.path.dirname(HERE), "iterative_solvers_amd", "host", "its_solve")


def parse(path, **env):
    out = subprocess.run([EXE, "--parse-only", path], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
    return out.returncode, out.stdout.split()[:4], out.stdout


@pytest.mark.parametrize("m", ["lap2d_16", "lap3d_8", "cd27_6", "pl_200"])
def test_fast_reader_equals_fscanf_reader_on_golden(m):
    assert os.path.exists(EXE), "build it: make -C iterative_solvers_amd/host"
    p = os.path.join(G, m + ".mtx")
    rc0, slow, _ = parse(p, ITS_MTX_READER="fscanf")
    rc1, fast, _ = parse(p)
    assert rc0 == 0 and rc1 == 0 and slow == fast and int(slow[2]) > 0


VARIANTS = {
    "symmetric_real": "%%MatrixMarket matrix coordinate real symmetric\n% comment\n%another\n4 4 5\n1 1 2.5\n3 1 -1e-3\n2 2 4\n4 2 1.25E+2\n4 4 .5\n",
    "pattern_general": "%%MatrixMarket matrix coordinate pattern general\n3 3 4\n1 1\n3 2\n2 2\n1 3\n",
    "pattern_symmetric": "%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n1 1\n3 1\n3 3\n",
    "integer_general": "%%MatrixMarket matrix coordinate integer general\n2 2 3\n2 1 7\n1 1 -3\n2 2 +4\n",
    "crlf_and_blanks": "%%MatrixMarket matrix coordinate real general\r\n2 2 3\r\n  1   1   1.0  \r\n2\t2\t2.0\r\n\r\n1 2 3.0000000000000004\r\n",
    "unsorted_rows": "%%MatrixMarket matrix coordinate real general\n3 3 6\n3 3 9\n1 2 1\n2 1 2\n1 1 3\n3 1 4\n2 2 5\n",
    "mixed_case_banner": "%%MatrixMarket MATRIX Coordinate Real General\n1 1 1\n1 1 0.1\n",
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_fast_reader_variants(tmp_path, name):
    p = tmp_path / (name + ".mtx")
    p.write_text(VARIANTS[name], newline="")
    rc0, slow, o0 = parse(str(p), ITS_MTX_READER="fscanf")
    rc1, fast, o1 = parse(str(p))
    assert rc0 == 0 and rc1 == 0, (o0, o1)
    assert slow == fast, (slow, fast)


def test_fast_reader_large_random(tmp_path):
    """> 1 MiB body: the multi-threaded path, random order, 17-digit values."""
    rng = np.random.default_rng(5)
    n, nz = 5000, 120_000
    I = rng.integers(1, n + 1, nz); J = rng.integers(1, n + 1, nz); V = rng.standard_normal(nz) * 10.0 ** rng.integers(-8, 8, nz)
    p = tmp_path / "rnd.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (n, n, nz))
        f.write("".join("%d %d %.17g\n" % t for t in zip(I, J, V)))
    assert os.path.getsize(p) > (1 << 20)
    _, slow, _ = parse(str(p), ITS_MTX_READER="fscanf")
    for threads in ("1", "3", "8"):
        _, fast, _ = parse(str(p), ITS_MTX_THREADS=threads)
        assert slow == fast, threads


def test_binary_cache_round_trip(tmp_path):
    src = os.path.join(G, "cd27_6.mtx")
    p = tmp_path / "m.mtx"
    p.write_bytes(open(src, "rb").read())
    _, want, _ = parse(str(p), ITS_MTX_READER="fscanf")
    _, first, _ = parse(str(p), ITS_MTX_CACHE="1")              # writes m.mtx.itsbin
    assert os.path.exists(str(p) + ".itsbin") and first == want
    _, second, _ = parse(str(p), ITS_MTX_CACHE="1")             # loads it
    assert second == want
    # a changed source invalidates the cache (size / mtime in the header)
    p.write_text(VARIANTS["unsorted_rows"])
    _, changed, _ = parse(str(p), ITS_MTX_CACHE="1")
    _, changed_ref, _ = parse(str(p), ITS_MTX_READER="fscanf")
    assert changed == changed_ref and changed != want


@pytest.mark.parametrize("body", [
    "%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n2 2 2.0\n",          # short body
    "%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1.0\n",                    # not square
    "%MatrixMarket matrix coordinate real general\n1 1 1\n1 1 1.0\n",                     # bad banner
    "%%MatrixMarket matrix array real general\n1 1\n1.0\n",                                # unsupported
])
def test_malformed_files_end_in_the_reference_reader(tmp_path, body):
    """The fast path declines; the fscanf reader reports exactly as the reference does (same text, same exit code)."""
    p = tmp_path / "bad.mtx"
    p.write_text(body)
    rc0, _, o0 = parse(str(p), ITS_MTX_READER="fscanf")
    rc1, _, o1 = parse(str(p))
    assert (rc0, o0) == (rc1, o1)
