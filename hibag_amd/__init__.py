"""hibag_amd -- MI355X-native implementation of HIBAG's attribute-bagging
prediction hot path (haplotype-pair posterior loop) behind the reference's own
interface: ``hlaSetKernelTarget("hip")``, ``hlaModelFromObj``, ``hlaPredict``.

The compute lives in ``csrc/libhibag_hip.so`` (hand-written HIP for gfx950,
C ABI in ``include/hibag_hip.h``); there is no CPU fallback in this package.
"""

from .model import (NA_INTEGER, Classifier, HlaAttrBagObj, HlaSNPGeno, engine_kind, engine_nkb, engine_steps, load_geno, load_model,  # noqa: F401
                    model_to_robj, save_model)
from .hibag import (HlaAlleleClass, HlaAttrBagClass, hlaClose, hlaModelFromObj, hlaModelToObj,   # noqa: F401
                    hlaPredict, hlaSetKernelTarget)
from .snpmatch import hlaGenoSwitchStrand, hlaSNPID  # noqa: F401
from .bed import HlaBEDGeno, hlaBED2Geno, hlaLociInfo  # noqa: F401
from .train import (RRandom, hlaAllele, hlaAttrBagging, hlaConcurrentAttrBagging, hlaParallelAttrBagging,  # noqa: F401
                    hlaUniqueAllele, set_seed)
from .merge import hlaAlleleDigit, hlaPredMerge  # noqa: F401
from .evaluate import (hlaAlleleSubset, hlaCompareAllele, hlaFlankingSNP, hlaGenoSubset, hlaSplitAllele,  # noqa: F401
                       r_sample)
from ._lib import HibagHipError  # noqa: F401

__all__ = ["engine_kind", "engine_nkb", "engine_steps", "NA_INTEGER", "Classifier", "HlaAttrBagObj", "HlaSNPGeno", "load_geno", "load_model", "model_to_robj", "save_model",
           "HlaAlleleClass", "HlaAttrBagClass", "hlaClose", "hlaModelFromObj", "hlaModelToObj",
           "hlaPredict", "hlaSetKernelTarget", "hlaGenoSwitchStrand", "hlaSNPID", "HibagHipError",
           "HlaBEDGeno", "hlaBED2Geno", "hlaLociInfo", "RRandom", "hlaAllele", "hlaAttrBagging", "hlaConcurrentAttrBagging", "hlaParallelAttrBagging", "hlaUniqueAllele", "hlaAlleleDigit", "hlaPredMerge", "hlaAlleleSubset", "hlaCompareAllele", "hlaFlankingSNP", "hlaGenoSubset",
           "hlaSplitAllele", "r_sample",
           "set_seed"]
