from .LightGCN import LightGCN  # noqa: F401
from .FREEDOM import FREEDOM  # noqa: F401
from .MMGCN import MMGCN  # noqa: F401
from .NGCF import NGCF  # noqa: F401
from .MGCN import MGCN  # noqa: F401
from .LayerGCN import LayerGCN  # noqa: F401
from .BPR import BPRMF  # noqa: F401
from .VBPR import VBPR  # noqa: F401
