"""Host-side mirror of the reference's interface for the hot path (witch_msa.gcmm).

The reference runs HMMER once per (HMM, chunk) and once per (query, HMM) from forked pool
workers.  HIP contexts do not survive fork(), so the MI355X path runs ONCE, batched, in
the parent process (engine.QueryAlignmentEngine) before the worker pool exists; the
functions below keep the reference's names, argument meaning, return types and error
behaviour and serve their answers from that precomputed table:

    rankBitscores   witch_msa/gcmm/loader.py:369-376
    writeWeights    witch_msa/gcmm/weighting.py:121-169
    getBackbones    witch_msa/gcmm/aligner.py:33-148
    search          witch_msa/gcmm/algorithm.py:273-336 (result files: :524-537)
    alignSubQueriesNew  witch_msa/gcmm/aligner.py:350-538 (weighted consensus DP on the GPU)
    mergeAlignmentsCollapsed  witch_msa/gcmm/merger.py:40-131 (final transitive merge, closed form)
    callback_queryAlignment / readCheckpointAlignments  witch_msa/gcmm/callback.py:9-29, loader.py:95-150 (checkpoint file)
    subset_alignment_and_hmmbuild / build_ehmm  witch_msa/gcmm/algorithm.py:394-477 (the model of a subset, without hmmbuild)

INTEGRATION.md shows the three-line change in witch_msa/gcmm/gcmm.py that installs them.
"""
from .engine import QueryAlignmentEngine, install, current_engine, warm_up  # noqa: F401
from .loader import rankBitscores, readAndRankBitscoreMP  # noqa: F401
from .weighting import writeWeights, calculateWeights, writeWeightsToLocal, readWeightsFromLocal  # noqa: F401
from .aligner import getBackbones  # noqa: F401
from .algorithm import search, check_query_names, divide_to_equal_chunks, num_chunks_for  # noqa: F401
from .merge import alignSubQueriesNew, compressInsertions, trace_to_string  # noqa: F401
from .merger import mergeAlignmentsCollapsed, mergeAlignmentsDevice  # noqa: F401
from .checkpoint import callback_queryAlignment, readCheckpointAlignments, writeCheckpointAlignments  # noqa: F401
from .hmmbuild import subset_alignment_and_hmmbuild, build_ehmm, hmmbuild_text  # noqa: F401
