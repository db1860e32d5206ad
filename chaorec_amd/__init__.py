"""chaorec_amd -- MI355X-native GCN-propagate + BPR + full-rank hot path behind ChaoRec's model surface.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels for gfx950 + the C-ABI (include/chaorec_hip.h)
  _lib.py          ctypes binding / build of libchaorec_hip.so
  graph.py         host-side graph setup (edge list -> CSR in HBM)
  ops.py           autograd-aware wrappers over the C-ABI
  Model/           LightGCN, MMGCN, FREEDOM with the reference's constructor / loss / gene_ranklist surface
  BasicGCN.py      the conv primitives MMGCN uses
  dataload.py, metrics.py, utils.py, arg_parser.py, train_and_evaluate.py, main.py
                   the reference's entry points for this path
  dist.py          user-row sharding over the GPUs of one node (RCCL)
"""
__version__ = "0.1.0"
