"""transcodon_amd: the TransCodon encoder forward and codon decode on MI355X (gfx950).

Host-side mirror of the reference's Python API (Alphabet, sequence classes,
ProteinBertModel and its sub-modules, embed_sequence) over hand-written HIP kernels in
``csrc/`` reached through a C ABI (``include/transcodon_hip.h``).
"""

from .alphabet import Alphabet, BatchConverter, ORGANISM2ID  # noqa: F401
from .model import GraphedForward, ProteinBertModel  # noqa: F401
from .modules import (ESM1bLayerNorm, RNA_2D_RobertaLMHead, RobertaLMHead,  # noqa: F401
                      TransformerLayer, gelu)
from .multihead_attention import MultiheadAttention  # noqa: F401
from .infer_utr import (AA_tokenize, embed_sequence, extract_embeddings, optimize_codons,  # noqa: F401
                        tokenize)
from .config import default_args, init_weights  # noqa: F401
