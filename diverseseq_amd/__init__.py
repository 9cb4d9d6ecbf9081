"""MI355X-native implementation of DiverseSeq's k-mer / delta-JSD / mash hot path.

``diverseseq_amd._dvs`` is the drop-in for the reference's ``diverse_seq._dvs``
extension module; ``diverseseq_amd.engine`` is the object layer over the C ABI
(include/dvs_hip.h); ``diverseseq_amd.distance`` the mash / euclidean drivers.
"""

__version__ = "0.1.0"
