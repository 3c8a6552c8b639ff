"""Level-0 drop-in: executables that WITCH can be pointed at through
its own plug-in keys `[Basic] hmmsearchpath` / `hmmalignpath` (witch_msa/default.config:15-17).

WITCH runs  "<hmmsearchpath> --cpu 1 --noali -E 99999999 -o OUT --max HMM FASTA"
(gcmm/algorithm.py:526-532) and "<hmmalignpath> -o OUT HMM FASTA" (gcmm/aligner.py:98-100)
through os.system from forked pool workers - one process per (HMM, chunk) and per
(query, HMM).  A HIP context cannot live in those short processes (and must not exist before a
fork), so the executables are thin C clients (client.c) of ONE resident server process
(server.py) that owns the GPU: they pass argv over a UNIX socket and exit with its status; the
server writes the output files - text that evalHMMSearchOutput (algorithm.py:579-605) and
ExtendedAlignment._read_sto (helpers/alignment_tools.py:926-955) accept.
"""
