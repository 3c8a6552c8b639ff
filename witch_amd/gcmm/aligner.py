"""getBackbones with the reference's signature (witch_msa/gcmm/aligner.py:33-148),
witch-ng branch (use_gcm=False): a lookup into the batched GPU alignment instead of one
hmmalign process per (query, HMM)."""
from .engine import current_engine


def getBackbones(index_to_hmm, taxon, taxon_ind, seq, query_path, sorted_weights,
                 workdir, backbone_dir, use_gcm=False):
    if use_gcm:
        raise NotImplementedError("witch_amd replaces the witch-ng path (use_gcm=False); the legacy "
                                  "GCM path writes per-HMM extended alignments for MAGUS and is out of scope")
    weights_map = {ind: w for (ind, w) in sorted_weights}
    if len(sorted_weights) == 0:
        return 'N/A', None                      # the reference's 2-tuple quirk (aligner.py:46-47)
    # adaptive inclusion until the cumulative weight reaches 0.999 (aligner.py:58-63)
    target = 0.999
    cur_sum = 0.
    idx = 0
    while idx < len(sorted_weights) and cur_sum < target:
        cur_sum += sorted_weights[idx][1]
        idx += 1
    top_k_hmms = [(w[0], float(w[1])) for w in sorted_weights[:idx]]
    ret_str = '{}\tpassed to main pipeline with top {} weights: {}'.format(
        taxon, len(top_k_hmms), top_k_hmms)     # aligner.py:66-67
    eng = current_engine()
    row = eng.taxon_row[taxon]
    subset_to_aligned_columns = dict()
    for i, _ in top_k_hmms:
        subset_to_aligned_columns[i] = eng.aligned_columns(row, i)   # aligner.py:126-142
    return ret_str, weights_map, subset_to_aligned_columns
