"""nextpolish2_amd — MI355X-native implementation of NextPolish2's per-contig consensus hot path.

Host-side mirror of the reference's interface for that path (src/main.rs:1819-1836) over the
C-ABI of include/np2.h; the compute lives in hand-written HIP kernels (csrc/)."""
from ._types import Opts, Pileup, Yak  # noqa: F401
from .api import BatchPolisher, Np2Error, Polisher, ResidentContig, fasta_record  # noqa: F401
