"""MI355X-native Partitioned Least Squares — host-side mirror of PartitionedLS.jl's interface for the hot path.

    from partls_amd import package; pls = package()
    model, _, report = pls.fit(pls.Opt, X, y, P, η=0.0)          # Opt.jl:73
    yhat = pls.predict(model, X)                                  # PartitionedLS.jl:152

Same names, argument meaning and result layout as the reference's exported API (PartitionedLS.jl:3):
fit, predict, PartLSFitResult, Opt, Alt, BnB, homogeneousCoords, regularizeProblem.
All arithmetic runs in libpartls_hip.so (hand-written HIP for gfx950) through the C ABI of include/partls.h;
there is no CPU fallback — importing works anywhere, computing needs an MI355X and raises otherwise.
"""
from .api import (Alt, BnB, Context, Frontier, IllConditionedWarning, MultiContext, Opt, PartLSFitResult, PartlsError, Report, build_library, default_context, default_multi, fit,
                  homogeneousCoords, library_path, predict, predict_device, regularizeProblem, synth_truth)
from . import _lib as lowlevel
from . import dist

__all__ = ["fit", "predict", "PartLSFitResult", "Opt", "Alt", "BnB", "homogeneousCoords", "regularizeProblem",
           "PartlsError", "IllConditionedWarning", "Report", "build_library", "library_path", "lowlevel", "Context", "Frontier", "MultiContext", "default_context", "default_multi", "synth_truth", "dist", "predict_device"]
