"""Level-0 shim, host side (no GPU): the text it writes is what the reference's parsers read,
the argv of WITCH's two command lines is understood, and the C clients talk to the server."""
import os
import subprocess
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "witch_amd", "shim", "bin")


class FakeBackend:
    """Deterministic stand-in for the GPU (NOT the oracle): score = 10 * length, columns = identity."""
    def search(self, hmm_path, records):
        from witch_amd.shim import formats
        hdr = formats.hmm_header(hmm_path)
        rows = [(n, 10.0 * len(t) + 0.5, 1.25, 1) for n, t in records if len(t) > 0 and not n.startswith("unrep")]
        return hdr, rows

    def align(self, jobs):
        out = []
        for hp, name, text in jobs:
            cols = np.arange(len(text), dtype=np.int32) + 2
            cols[:1] = -1
            out.append((len(text) + 5, cols))
        return out


def _hmm(tmp_path):
    p = tmp_path / "m.hmm"
    p.write_text("HMMER3/f [3.1b2 | February 2015]\nNAME  A_0_7\nLENG  12\nALPH  DNA\n"
                 "STATS LOCAL MSV      -9.0 0.71\nSTATS LOCAL VITERBI -9.5 0.71\nSTATS LOCAL FORWARD  -4.2 0.71\nHMM  A C G T\n")
    return str(p)


@pytest.fixture()
def server(tmp_path):
    from witch_amd.shim.server import Server
    sock = str(tmp_path / "s.sock")
    srv = Server(FakeBackend(), sock)
    ready = threading.Event()
    threading.Thread(target=srv.serve_forever, args=(ready,), daemon=True).start()
    assert ready.wait(10)
    return sock


def test_hmmsearch_table_is_readable_by_the_reference_parser(tmp_path):
    from witch_amd.shim import formats
    from tests.refparse import evalHMMSearchOutput
    hdr = formats.hmm_header(_hmm(tmp_path))
    assert hdr == {"name": "A_0_7", "M": 12, "ftau": -4.2, "flambda": 0.71}
    rows = [("q1", 123.4, 0.3, 1), ("a_long_query_name_with_many_chars", -5.2, 0.0, 2), ("q3", 7.0, 11.1, 1)]
    out = tmp_path / "o.txt"
    out.write_text(formats.format_hmmsearch("m.hmm", "q.fa", hdr, rows, 3))
    got = evalHMMSearchOutput(str(out))
    assert {k: v[1] for k, v in got.items()} == {"q1": 123.4, "a_long_query_name_with_many_chars": -5.2, "q3": 7.0}
    assert got["q1"][0] == 0.0 or got["q1"][0] < 1e-30          # E-value column parses as a float
    out.write_text(formats.format_hmmsearch("m.hmm", "q.fa", hdr, [], 3))
    assert evalHMMSearchOutput(str(out)) == {}


def test_stockholm_row_round_trip_against_hmmer_columns(golden_case):
    """cols -> Stockholm row -> the reference's decoding (aligner.py:126-142) -> cols, for every
    golden hmmalign result (inserts, flanks, deletions as HMMER placed them)."""
    from witch_amd.shim import formats
    n = 0
    for qn, a in golden_case.g["align"].items():
        seq = dict(zip(golden_case.qnames, golden_case.qseqs))[qn]
        for idx, cols in a["cols"].items():
            M = max([c for c in cols if c >= 0] + [0]) + 3
            row = formats.stockholm_row(seq, cols, M)
            assert len(row.replace("-", "")) == len(seq)
            assert sum(1 for ch in row if not ch.islower()) == M
            assert formats.decode_stockholm_row(row) == list(cols)
            sto = formats.format_stockholm(qn, row, width=60)
            body = "".join(l.split()[1] for l in sto.splitlines() if l and not l.startswith("#") and l != "//")
            assert body == row
            n += 1
    assert n > 0


def test_argv_of_witch_command_lines():
    from witch_amd.shim.server import parse_hmmsearch_argv, parse_hmmalign_argv, ArgError
    opts, hmm, fa = parse_hmmsearch_argv("--cpu 1 --noali -E 99999999 -o out.txt --max m.hmm q.fa".split())
    assert (hmm, fa, opts["-o"], opts["--cpu"], opts["-E"]) == ("m.hmm", "q.fa", "out.txt", "1", "99999999")
    assert opts["--max"] is True and opts["--noali"] is True
    opts, hmm, fa = parse_hmmalign_argv("-o a.sto m.hmm q.fa".split())
    assert (hmm, fa, opts["-o"]) == ("m.hmm", "q.fa", "a.sto")
    with pytest.raises(ArgError):
        parse_hmmsearch_argv(["m.hmm"])
    with pytest.raises(ArgError):
        parse_hmmalign_argv(["-o"])


def test_clients_end_to_end_with_a_stub_backend(server, tmp_path):
    if not os.path.exists(os.path.join(BIN, "hmmsearch")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "witch_amd", "shim")], check=True, stdout=subprocess.DEVNULL)
    from tests.refparse import evalHMMSearchOutput
    from witch_amd.shim import formats
    hmm = _hmm(tmp_path)
    fa = tmp_path / "q.fa"
    fa.write_text(">q1 some description\nACGTAC\nGT\n>unrep2\nAC\n>q3\nACG\n")
    env = dict(os.environ, WITCH_HIP_SOCKET=server)
    # the exact command line of gcmm/algorithm.py:526-532, relative paths resolved against the client's cwd
    r = subprocess.run([os.path.join(BIN, "hmmsearch"), "--cpu", "1", "--noali", "-E", "99999999", "-o", "res.txt",
                        "--max", "m.hmm", "q.fa"], cwd=str(tmp_path), env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = evalHMMSearchOutput(str(tmp_path / "res.txt"))
    assert {k: v[1] for k, v in got.items()} == {"q1": 80.5, "q3": 30.5}
    # gcmm/aligner.py:98-100
    one = tmp_path / "one.fa"
    one.write_text(">q1\nACGTACGT\n")
    r = subprocess.run([os.path.join(BIN, "hmmalign"), "-o", str(tmp_path / "a.sto"), hmm, str(one)], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in open(tmp_path / "a.sto") if l.strip() and not l.startswith("#") and l.strip() != "//"]
    row = "".join(x[1] for x in rows if x[0] == "q1")
    assert formats.decode_stockholm_row(row) == [-1] + list(range(3, 10))
    # failures exit non-zero with a message, and leave no output file
    r = subprocess.run([os.path.join(BIN, "hmmsearch"), "-o", str(tmp_path / "x.txt"), str(tmp_path / "missing.hmm"), str(fa)],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "HMM file" in r.stderr and not os.path.exists(tmp_path / "x.txt")
    r = subprocess.run([os.path.join(BIN, "hmmalign"), hmm, str(fa)], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "one sequence per call" in r.stderr
    # many concurrent hmmalign calls are batched and all answered
    procs = [subprocess.Popen([os.path.join(BIN, "hmmalign"), "-o", str(tmp_path / ("b%d.sto" % i)), hmm, str(one)], env=env)
             for i in range(24)]
    assert all(p.wait() == 0 for p in procs)
    assert all(os.path.getsize(tmp_path / ("b%d.sto" % i)) > 0 for i in range(24))
    from witch_amd.shim.server import request
    st, body = request(server, "ping", [])
    assert st == 0 and body.split()[:1] == ["pong"] and int(body.split()[2]) >= 25


def test_evalues_follow_hmmer_formula(golden_case):
    """E = Z * exp(-lambda (s - tau)) with the FORWARD line of the HMM file reproduces the E-values
    HMMER printed for the golden searches (to the 0.1-bit rounding of the printed score)."""
    from witch_amd.shim import formats
    case = golden_case
    Z = len(case.qnames)
    n = 0
    for hf, hp in zip(case.hmm_files[:3], case.hmm_paths[:3]):
        hdr = formats.hmm_header(hp)
        if hdr["flambda"] is None:
            continue
        for q, v in case.g["search"][hf].items():
            ev = formats.forward_evalue(v["score"], Z, hdr["ftau"], hdr["flambda"])
            if v["evalue"] > 1e-250 and ev > 1e-250 and v["score"] > hdr["ftau"] + 1:
                assert 0.93 < ev / v["evalue"] < 1.08, (hf, q, ev, v["evalue"])
                n += 1
    assert n > 0 or case.name.startswith("example")


def test_only_one_server_survives_a_stampede(tmp_path):
    """WITCH fires num_cpus hmmsearch calls at once; every client that cannot connect launches a
    server.  The flock on <socket>.lock (taken before the socket or the GPU is touched) leaves exactly
    one; the socket lives in a directory private to the user."""
    import sys
    import time
    from witch_amd.shim.server import request
    d = tmp_path / "rt"
    sock = str(d / "witch_hip" / "server.sock")
    env = dict(os.environ, PYTHONPATH=ROOT, WITCH_HIP_IDLE_TIMEOUT="30")
    procs = [subprocess.Popen([sys.executable, "-m", "witch_amd.shim.server", "--socket", sock], env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(8)]
    deadline = time.time() + 60
    while time.time() < deadline and sum(p.poll() is None for p in procs) > 1:
        time.sleep(0.2)
    alive = [p for p in procs if p.poll() is None]
    assert len(alive) == 1, len(alive)
    assert all(p.returncode == 0 for p in procs if p.poll() is not None)     # the losers leave quietly
    for _ in range(50):
        if os.path.exists(sock):
            break
        time.sleep(0.1)
    st, body = request(sock, "ping", [])
    assert st == 0 and body.startswith("pong")
    assert (os.stat(os.path.dirname(sock)).st_mode & 0o077) == 0
    st, _ = request(sock, "shutdown", [])
    assert alive[0].wait(10) == 0


def test_model_cache_is_bounded_and_options_are_checked(tmp_path, monkeypatch):
    from witch_amd.shim import server as srv

    class _H:
        closed = 0

        def __init__(self, *a, **k):
            pass

        def close(self):
            _H.closed += 1
    import witch_amd.ehmm as ehmm_mod
    monkeypatch.setattr(ehmm_mod, "EHMM", _H)
    b = srv.GpuBackend(0, max_models=3)
    paths = []
    for i in range(5):
        p = tmp_path / ("m%d.hmm" % i)
        p.write_text(open(_hmm(tmp_path)).read())
        paths.append(str(p))
        b.model(str(p))
    assert len(b.cache) == 3 and _H.closed == 2 and list(b.cache) == [os.path.realpath(x) for x in paths[2:]]
    b.model(paths[2])                                   # touching a model makes it the most recent one
    b.model(paths[0])
    assert os.path.realpath(paths[3]) not in b.cache and os.path.realpath(paths[2]) in b.cache
    # WITCH's own command line passes; anything that would change the reported set in HMMER is refused
    ok = srv.parse_hmmsearch_argv("--cpu 1 --noali -E 99999999 -o o --max m q".split())[0]
    srv.check_hmmsearch_options(ok)
    for bad in ("--cpu 1 --noali -E 99999999 -o o m q", "--max -E 10 m q", "--max -T 5 m q", "--max --nonull2 m q"):
        with pytest.raises(srv.ArgError):
            srv.check_hmmsearch_options(srv.parse_hmmsearch_argv(bad.split())[0])
