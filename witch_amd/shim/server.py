"""Resident GPU server behind the level-0 hmmsearch / hmmalign executables (see __init__.py).

    python -m witch_amd.shim.server --socket /tmp/witch_hip.sock [--device 0] [--daemonize]

Protocol (one request per connection): the client sends  tool \\0 cwd \\0 arg \\0 arg ... and shuts
its write side; the server answers "<exit status>\\n<message>".  Requests are served by one
thread per connection; GPU work is serialised, and hmmalign requests that arrive together (WITCH
issues them from up to num_cpus workers at once) are batched into ONE wh_align launch.
"""
import argparse
import os
import socket
import sys
import threading
import time

import numpy as np

from . import formats


def default_socket_path():
    """$WITCH_HIP_SOCKET, else a socket inside a directory only this user can enter
    ($XDG_RUNTIME_DIR/witch_hip or /tmp/witch_hip_<uid>, mode 0700; client.c uses the same rule)."""
    if os.environ.get("WITCH_HIP_SOCKET"):
        return os.environ["WITCH_HIP_SOCKET"]
    base = os.environ.get("XDG_RUNTIME_DIR")
    d = os.path.join(base, "witch_hip") if base else "/tmp/witch_hip_%d" % os.getuid()
    return os.path.join(d, "server.sock")


def private_socket_dir(sock_path):
    """Create the socket's directory 0700 and refuse one that somebody else owns or can write to."""
    d = os.path.dirname(os.path.abspath(sock_path))
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if d.startswith("/tmp/witch_hip_") or d.endswith("/witch_hip"):
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError("socket directory %s is not private to uid %d" % (d, os.getuid()))
    return d


class ArgError(Exception):
    pass


def parse_hmmsearch_argv(argv):
    """The options WITCH passes (algorithm.py:526-532) plus the ones that take a value in HMMER,
    so that positional arguments are found; unknown flags are ignored like a no-op."""
    takes_value = {"-o", "-A", "-E", "-T", "-Z", "--cpu", "--tblout", "--domtblout", "--pfamtblout", "--domE", "--domT",
                   "--incE", "--incT", "--incdomE", "--incdomT", "--F1", "--F2", "--F3", "--domZ", "--seed", "--tformat",
                   "--textw"}
    opts, pos, i = {}, [], 0
    while i < len(argv):
        a = argv[i]
        if a in takes_value:
            if i + 1 >= len(argv):
                raise ArgError("option %s needs a value" % a)
            opts[a] = argv[i + 1]
            i += 2
        elif a.startswith("-") and a != "-":
            opts[a] = True
            i += 1
        else:
            pos.append(a)
            i += 1
    if len(pos) != 2:
        raise ArgError("Incorrect number of command line arguments.\nUsage: hmmsearch [options] <hmmfile> <seqdb>")
    return opts, pos[0], pos[1]


def check_hmmsearch_options(opts):
    """The GPU path computes what WITCH's own command line asks for (algorithm.py:526-532):
    `--max` (no filters) with a reporting threshold that lets every hit through.  Anything that would
    change the reported set in HMMER (filters on, a real E-value / score cut-off, no null2) is refused
    with exit status 1 instead of being silently ignored."""
    if "--max" not in opts:
        raise ArgError("witch-hip hmmsearch shim: only the unfiltered search (--max) is implemented; "
                       "WITCH passes --max unless its filters are switched on")
    for o in ("-T", "--incT", "--domT", "--incdomT", "--nonull2", "--nobias", "--cut_ga", "--cut_nc", "--cut_tc"):
        if o in opts:
            raise ArgError("witch-hip hmmsearch shim: option %s is not supported" % o)
    for o in ("-E", "--incE", "--domE", "--incdomE"):
        if o in opts:
            try:
                v = float(opts[o])
            except ValueError:
                raise ArgError("witch-hip hmmsearch shim: bad value for %s" % o)
            if o == "-E" and v < 1e6:
                raise ArgError("witch-hip hmmsearch shim: -E %s would drop hits; only E >= 1e6 (WITCH: 99999999) "
                               "is supported, every sequence with a domain is reported" % opts[o])


def parse_hmmalign_argv(argv):
    takes_value = {"-o", "--mapali", "--informat", "--outformat"}
    opts, pos, i = {}, [], 0
    while i < len(argv):
        a = argv[i]
        if a in takes_value:
            if i + 1 >= len(argv):
                raise ArgError("option %s needs a value" % a)
            opts[a] = argv[i + 1]
            i += 2
        elif a.startswith("-") and a != "-":
            opts[a] = True
            i += 1
        else:
            pos.append(a)
            i += 1
    if len(pos) != 2:
        raise ArgError("Incorrect number of command line arguments.\nUsage: hmmalign [-options] <hmmfile> <seqfile>")
    return opts, pos[0], pos[1]


class GpuBackend:
    """The only place that touches libwitch_hip.so.  One single-model EHMM per HMM file
    (hmmsearch works one model at a time; hmmalign batches are grouped by model)."""

    def __init__(self, device=0, max_models=None):
        import collections
        self.device = device
        # realpath -> (mtime, EHMM, header), least recently used first.  Every handle owns device
        # workspace (hundreds of MB once it has scored a chunk) and a WITCH ensemble has hundreds of
        # models, so the cache is bounded and evicted handles are closed.
        self.cache = collections.OrderedDict()
        self.max_models = max_models or int(os.environ.get("WITCH_HIP_MAX_MODELS", "32"))
        self.lock = threading.Lock()

    def model(self, path):
        from witch_amd.ehmm import EHMM
        rp = os.path.realpath(path)
        mt = os.stat(rp).st_mtime
        hit = self.cache.get(rp)
        if hit is None or hit[0] != mt:
            if hit is not None:
                hit[1].close()
                del self.cache[rp]
            while len(self.cache) >= self.max_models:
                _, (_, old, _) = self.cache.popitem(last=False)
                old.close()
            hit = (mt, EHMM([rp], device=self.device), formats.hmm_header(rp))
            self.cache[rp] = hit
        else:
            self.cache.move_to_end(rp)
        return hit[1], hit[2]

    def search(self, hmm_path, records):
        """records: [(name, text)] -> [(name, bits, bias_bits, n_domains)] of reported sequences."""
        from witch_amd.ehmm import pack_queries
        with self.lock:
            e, hdr = self.model(hmm_path)
            seqs = [e.digitize(t.upper()) for _, t in records]     # alignment_tools.py:730-731 upper-cases
            res, offs = pack_queries(seqs)
            deci, flags, det = e.score(res, offs, want_detail=True)
        rows = []
        for i, (name, _) in enumerate(records):
            if flags[i, 0] & 1:
                d = det[i]
                bias = max(0.0, float(d.pre_score) - float(d.seq_score))
                rows.append((name, deci[i, 0] / 10.0, bias, int(d.nenv)))
        return hdr, rows

    def align(self, jobs):
        """jobs: [(hmm_path, name, text)] -> [(M, cols)]; one launch per distinct model."""
        from witch_amd.ehmm import pack_queries
        out = [None] * len(jobs)
        by_model = {}
        for j, (hp, _, _) in enumerate(jobs):
            by_model.setdefault(os.path.realpath(hp), []).append(j)
        with self.lock:
            for hp, idxs in by_model.items():
                e, hdr = self.model(hp)
                seqs = [e.digitize(jobs[j][2].upper()) for j in idxs]
                res, offs = pack_queries(seqs)
                cols, co = e.align(res, offs, np.arange(len(idxs)), np.zeros(len(idxs), dtype=np.int32))
                for n, j in enumerate(idxs):
                    out[j] = (int(e.M[0]), cols[co[n]:co[n + 1]].copy())
        return out


class AlignBatcher:
    """Collects hmmalign jobs that arrive within a short window and runs them together."""

    def __init__(self, backend, window_s=0.0005, max_batch=4096):
        self.backend, self.window, self.max_batch = backend, window_s, max_batch
        self.cv = threading.Condition()
        self.queue = []
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def submit(self, job):
        slot = {"job": job, "done": threading.Event(), "result": None, "error": None}
        with self.cv:
            self.queue.append(slot)
            self.cv.notify()
        slot["done"].wait()
        if slot["error"] is not None:
            raise slot["error"]
        return slot["result"]

    def _run(self):
        while True:
            with self.cv:
                while not self.queue:
                    self.cv.wait()
            time.sleep(self.window)                       # let the other workers' requests arrive
            with self.cv:
                batch, self.queue = self.queue[:self.max_batch], self.queue[self.max_batch:]
            try:
                res = self.backend.align([s["job"] for s in batch])
                for s, r in zip(batch, res):
                    s["result"] = r
            except Exception as ex:                       # one bad job fails alone on the retry
                for s in batch:
                    try:
                        s["result"] = self.backend.align([s["job"]])[0]
                    except Exception as ex1:
                        s["error"] = ex1
                del ex
            for s in batch:
                s["done"].set()


class Server:
    def __init__(self, backend, sock_path):
        self.backend = backend
        self.batcher = AlignBatcher(backend)
        self.sock_path = sock_path
        self.stats = {"hmmsearch": 0, "hmmalign": 0}

    # ---------------------------------------------------------------- the two tools
    def run_hmmsearch(self, argv, cwd):
        opts, hmm, fa = parse_hmmsearch_argv(argv)
        hmm, fa = os.path.join(cwd, hmm), os.path.join(cwd, fa)
        if not os.path.exists(hmm):
            raise ArgError("Error: File existence/permissions problem in trying to open HMM file %s." % hmm)
        if not os.path.exists(fa):
            raise ArgError("Error: Failed to open sequence file %s for reading" % fa)
        check_hmmsearch_options(opts)
        records = formats.read_fasta(fa)
        hdr, rows = self.backend.search(hmm, records)
        text = formats.format_hmmsearch(hmm, fa, hdr, rows, len(records))
        if "-o" in opts:
            with open(os.path.join(cwd, opts["-o"]), "w") as f:
                f.write(text)
            return ""
        return text

    def run_hmmalign(self, argv, cwd):
        opts, hmm, fa = parse_hmmalign_argv(argv)
        hmm, fa = os.path.join(cwd, hmm), os.path.join(cwd, fa)
        if not os.path.exists(hmm):
            raise ArgError("Error: File existence/permissions problem in trying to open HMM file %s." % hmm)
        if not os.path.exists(fa):
            raise ArgError("Error: Failed to open sequence file %s for reading" % fa)
        records = formats.read_fasta(fa)
        if len(records) != 1:
            # WITCH aligns one query per call (aligner.py:90-100); a multi-sequence alignment
            # needs the shared insert columns hmmalign computes, which this shim does not emit
            raise ArgError("witch-hip hmmalign shim: exactly one sequence per call (got %d)" % len(records))
        name, text = records[0]
        M, cols = self.batcher.submit((hmm, name, text))
        sto = formats.format_stockholm(name, formats.stockholm_row(text, cols, M))
        if "-o" in opts:
            with open(os.path.join(cwd, opts["-o"]), "w") as f:
                f.write(sto)
            return ""
        return sto

    # ---------------------------------------------------------------- plumbing
    def handle(self, conn):
        try:
            buf = b""
            while True:
                chunk = conn.recv(65536)
                if not chunk:
                    break
                buf += chunk
            parts = buf.split(b"\0")
            if parts and parts[-1] == b"":
                parts.pop()
            if len(parts) < 2:
                raise ArgError("malformed request")
            tool, cwd, argv = parts[0].decode(), parts[1].decode(), [p.decode() for p in parts[2:]]
            if tool == "hmmsearch":
                out = self.run_hmmsearch(argv, cwd)
            elif tool == "hmmalign":
                out = self.run_hmmalign(argv, cwd)
            elif tool == "shutdown":
                conn.sendall(b"0\nbye\n")
                conn.close()
                os._exit(0)
            elif tool == "ping":
                out = "pong %d %d\n" % (self.stats["hmmsearch"], self.stats["hmmalign"])
            else:
                raise ArgError("unknown tool %r" % tool)
            if tool in self.stats:
                self.stats[tool] += 1
            conn.sendall(b"0\n" + out.encode())
        except ArgError as ex:
            conn.sendall(b"1\n" + str(ex).encode() + b"\n")
        except Exception as ex:                           # HMMER exits 1 with a message; so do we
            conn.sendall(b"1\nError: " + ("%s: %s" % (type(ex).__name__, ex)).encode() + b"\n")
        finally:
            try:
                conn.close()
            except OSError:
                pass

    def serve_forever(self, ready=None, idle_timeout=None):
        """One server per socket: an exclusive flock on <socket>.lock is taken BEFORE the socket is
        touched (and before any GPU call), so of the many clients WITCH starts at once - each of which
        launches a server when it cannot connect - exactly one server survives; the others exit 0 and
        their clients connect to the survivor.  The server leaves after <idle_timeout> seconds without
        a request ($WITCH_HIP_IDLE_TIMEOUT, default 3600; 0 = never)."""
        import fcntl
        private_socket_dir(self.sock_path)
        self._lockf = open(self.sock_path + ".lock", "w")
        for attempt in range(100):
            try:
                fcntl.flock(self._lockf, fcntl.LOCK_EX | fcntl.LOCK_NB)
                break
            except OSError:
                # another server owns this socket - unless it is just leaving (idle shutdown: path already
                # unlinked, lock released a moment later): then wait for the lock instead of giving up
                if os.path.exists(self.sock_path) or attempt == 99:
                    return False
                time.sleep(0.05)
        if os.path.exists(self.sock_path):
            os.unlink(self.sock_path)
        old_umask = os.umask(0o077)
        try:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.sock_path)
        finally:
            os.umask(old_umask)
        srv.listen(512)
        if idle_timeout is None:
            idle_timeout = float(os.environ.get("WITCH_HIP_IDLE_TIMEOUT", "3600"))
        srv.settimeout(idle_timeout if idle_timeout > 0 else None)
        if ready is not None:
            ready.set()
        while True:
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                if threading.active_count() > 2:          # requests still running: keep going
                    continue
                # Leave without dropping anyone: unlink the path FIRST (a client arriving from now on finds no
                # socket and starts a fresh server, which waits on the flock until this one is gone), then serve
                # whatever is already in the listen backlog before closing.
                try:
                    os.unlink(self.sock_path)
                except OSError:
                    pass
                fcntl.flock(self._lockf, fcntl.LOCK_UN)    # a successor may bind a new socket while the backlog drains
                srv.setblocking(False)
                late = []
                while True:
                    try:
                        conn, _ = srv.accept()
                    except (BlockingIOError, OSError):
                        break
                    conn.settimeout(None)
                    t = threading.Thread(target=self.handle, args=(conn,), daemon=True)
                    t.start()
                    late.append(t)
                for t in late:
                    t.join()
                srv.close()
                return True
            conn.settimeout(None)
            threading.Thread(target=self.handle, args=(conn,), daemon=True).start()


def request(sock_path, tool, argv, cwd=None):
    """Python twin of client.c (tests and tooling): returns (status, text)."""
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(sock_path)
    s.sendall(b"\0".join([tool.encode(), (cwd or os.getcwd()).encode()] + [a.encode() for a in argv]) + b"\0")
    s.shutdown(socket.SHUT_WR)
    buf = b""
    while True:
        chunk = s.recv(65536)
        if not chunk:
            break
        buf += chunk
    s.close()
    head, _, body = buf.partition(b"\n")
    return int(head or b"1"), body.decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--socket", default=default_socket_path())
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--daemonize", action="store_true", help="detach (the clients start the server this way)")
    args = ap.parse_args()
    if args.daemonize:
        # fork BEFORE anything touches HIP
        if os.fork() > 0:
            return
        os.setsid()
        if os.fork() > 0:
            os._exit(0)
        private_socket_dir(args.socket)
        log = open(args.socket + ".log", "a")
        os.dup2(log.fileno(), 1)
        os.dup2(log.fileno(), 2)
        sys.stdin.close()
    private_socket_dir(args.socket)
    Server(GpuBackend(args.device), args.socket).serve_forever()


if __name__ == "__main__":
    main()
