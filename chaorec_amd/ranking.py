"""The one `gene_ranklist` behind the 49 near-identical copies in the reference's Model/*.py:
score = user @ item.T, history -> mask_value (1e-6 in 46 models, 1e-5 in 3), top-k, + num_user, LongTensor on CPU.

    from chaorec_amd.ranking import gene_ranklist, history_csr
    self.hist = history_csr(user_item_dict, num_user, device)          # once, in __init__
    return gene_ranklist(self.result, self.num_user, self.num_item, self.hist, 1e-6, topk)
"""
import torch

from . import graph, ops


def history_csr(user_item_dict, num_user, device):
    rowptr, col = graph.user_hist_csr(user_item_dict, num_user)
    return rowptr.to(device), col.to(device)


def _to_host(idx):
    """[U, K] int64 rank list -> CPU LongTensor through page-locked memory (torch's caching host allocator hands the
    same pages out again call after call; a pageable .cpu() copy of sports' 11.6 MB list costs several times the
    ranking itself)."""
    buf = torch.empty(idx.shape, dtype=idx.dtype, pin_memory=True)
    buf.copy_(idx, non_blocking=True)
    torch.cuda.current_stream(idx.device).synchronize()
    return buf


def gene_ranklist(result, num_user, num_item, hist, mask_value=1e-6, topk=50, to_cpu=True):
    """result [N, D] (users first) on the GPU -> LongTensor [num_user, topk] of GLOBAL item ids on the CPU (the
    reference's contract); to_cpu=False keeps it in HBM for utils.gene_metrics_device."""
    with torch.no_grad():
        result = result.detach()
        idx, _ = ops.score_topk(result[:num_user], result[num_user:num_user + num_item], hist, mask_value, topk,
                                id_offset=num_user)
    return _to_host(idx) if to_cpu else idx
