"""Text formats of the level-0 shim: FASTA in, hmmsearch per-sequence table and single-sequence
Stockholm out.  Pure Python, no GPU."""
import math


def read_fasta(path):
    """[(name, sequence)] - name is the first word of the header, like HMMER's sqio."""
    out, name, chunks = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(chunks)))
                name = line[1:].split()[0] if len(line) > 1 and line[1:].split() else ""
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        out.append((name, "".join(chunks)))
    return out


def hmm_header(path):
    """NAME, LENG and the Forward E-value parameters (tau, lambda) of a HMMER3/f file."""
    info = {"name": "unnamed", "M": 0, "ftau": None, "flambda": None}
    opener = open
    if str(path).endswith(".gz"):
        import gzip
        opener = gzip.open
    with opener(path, "rt") as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "NAME":
                info["name"] = w[1]
            elif w[0] == "LENG":
                info["M"] = int(w[1])
            elif w[0] == "STATS" and len(w) >= 5 and w[2] == "FORWARD":
                info["ftau"], info["flambda"] = float(w[3]), float(w[4])
            elif w[0] == "HMM":
                break
    return info


def forward_evalue(bits, n_targets, ftau, flambda):
    """E = Z * P(score >= s) with the exponential tail HMMER fits for Forward scores
    (P = exp(-lambda (s - tau)) for s >= tau, else 1).  WITCH never reads it (loader.py:293)."""
    if ftau is None or flambda is None:
        return 0.0
    x = bits - ftau
    p = 1.0 if x < 0 else math.exp(max(-745.0, -flambda * x))
    return p * n_targets


def _g(x):
    return "%9.2g" % x


def format_hmmsearch(hmm_path, fasta_path, hdr, rows, n_targets):
    """rows: [(name, bits, bias_bits, n_domains)] of the REPORTED sequences.  The table is what
    evalHMMSearchOutput reads: a line starting with 'E-value', then >= 9 whitespace-separated
    fields per row (E-value, score, bias, best-domain E-value/score/bias, exp, N, name), a blank
    line to end.  The best-domain columns repeat the full-sequence values (not computed)."""
    namew = max([8] + [len(r[0]) for r in rows])
    out = []
    out.append("# hmmsearch :: search profile(s) against a sequence database")
    out.append("# witch-hip level-0 shim (MI355X); table layout of HMMER 3.1b2")
    out.append("# - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - -")
    out.append("# query HMM file:                  %s" % hmm_path)
    out.append("# target sequence database:        %s" % fasta_path)
    out.append("# - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - - -")
    out.append("")
    out.append("Query:       %s  [M=%d]" % (hdr["name"], hdr["M"]))
    out.append("Scores for complete sequences (score includes all domains):")
    out.append("   --- full sequence ---   --- best 1 domain ---    -#dom-")
    out.append("    E-value  score  bias    E-value  score  bias    exp  N  %-*s Description" % (namew, "Sequence"))
    out.append("    ------- ------ -----    ------- ------ -----   ---- --  %s -----------" % ("-" * namew))
    if not rows:
        out.append("")
        out.append("   [No hits detected that satisfy reporting thresholds]")
    for name, bits, bias, ndom in sorted(rows, key=lambda r: -r[1]):
        ev = forward_evalue(bits, n_targets, hdr["ftau"], hdr["flambda"])
        out.append("  %s %6.1f %5.1f  %s %6.1f %5.1f  %5.1f %2d  %-*s " %
                   (_g(ev), bits, bias, _g(ev), bits, bias, float(max(ndom, 1)), max(ndom, 1), namew, name))
    out.append("")
    out.append("")
    out.append("Internal pipeline statistics summary:")
    out.append("-------------------------------------")
    out.append("Query model(s):                            1  (%d nodes)" % hdr["M"])
    out.append("Target sequences:                   %8d" % n_targets)
    out.append("//")
    out.append("[ok]")
    return "\n".join(out) + "\n"


def stockholm_row(seq_text, cols, M):
    """One hmmalign row: match columns 0..M-1 in order (uppercase residue or '-'), residues that
    are not in a match column (cols == -1: flanks and inserts) lowercase where they occur."""
    parts, c = [], 0
    for ch, col in zip(seq_text, cols):
        col = int(col)
        if col >= 0:
            if col > c:
                parts.append("-" * (col - c))
            parts.append(ch.upper())
            c = col + 1
        else:
            parts.append(ch.lower())
    if M > c:
        parts.append("-" * (M - c))
    return "".join(parts)


def format_stockholm(name, row, width=200):
    """Interleaved Stockholm blocks of <width> columns like hmmalign writes."""
    out = ["# STOCKHOLM 1.0", ""]
    namew = max(len(name), 1)
    for a in range(0, max(len(row), 1), width):
        out.append("%-*s %s" % (namew, name, row[a:a + width]))
        out.append("")
    out.append("//")
    return "\n".join(out) + "\n"


def decode_stockholm_row(row):
    """aligner.py:126-142: per residue the 0-based match column or -1."""
    cols, regular = [], 0
    for ch in row:
        if ch == '-':
            regular += 1
        elif ch == '.':
            continue
        elif ch.islower():
            cols.append(-1)
        else:
            cols.append(regular)
            regular += 1
    return cols
