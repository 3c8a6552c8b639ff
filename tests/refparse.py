"""Test-side readers of HMMER's text formats with the reference's contract (not product code:
the GPU path never parses text; these prove that the text the level-0 executables emit is what
the reference's parsers accept)."""
import re


def evalHMMSearchOutput(path):
    """The per-sequence table of hmmsearch as witch_msa/gcmm/algorithm.py:579-605 reads it:
    rows after the line starting with 'E-value' up to the first blank line, nine whitespace
    fields, rows containing '--' skipped; -> {name: (E-value, score)}."""
    nine = re.compile(r"\s+".join([r"(\S+)"] * 9))
    out, inside = {}, False
    for raw in open(path):
        line = raw.strip()
        if not inside:
            inside = line.startswith("E-value")
            continue
        if line == "":
            break
        m = nine.search(line)
        if m and "--" not in m.group(0):
            out[m.group(9).strip()] = (float(m.group(1)), float(m.group(2)))
    return out
