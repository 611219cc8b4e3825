from .IVFPQTopk import IVFPQTopk
from .Topk import Topk
