"""`longvgen`-compatible import aliases, so the reference's entry script keeps its import lines
(infer_cogvideo_mp_fifo.py:63-71):

    import tokensgen_amd.compat; tokensgen_amd.compat.install_longvgen_alias()
    from longvgen.models import CogVideoXTransformer3DModel            # -> tokensgen_amd.transformer
    from longvgen.schedulers import CogVideoXDPMScheduler              # -> tokensgen_amd.scheduler
    from longvgen.fifo_sampling import cogvideo_fifo_mp_v2             # -> tokensgen_amd.fifo
    from longvgen.pipeline import MPFIFOVideoIPAdapterCogVideoXPipeline, LongVGenCogVideoXPipeline
    from longvgen.video_ipadapter import Resampler                     # -> tokensgen_amd.resampler

A top-level module `pca` (class PCA) is registered as well, because the reference pickles its `pca.PCA` object whole and
`torch.load`s it in the T2To pipeline (pipeline_cogvideox_t2to.py:771).  Only the hot-path names exist; anything else of the
reference package (data loading, training utilities) is deliberately absent and raises ImportError."""
import sys
import types


def ensure_pca_module():
    """Register tokensgen_amd.pca as the top-level module `pca` unless one is importable already (the reference pickles `pca.PCA` objects whole:
    resampler.py:207, pipeline_cogvideox_t2to.py:698,771)."""
    if "pca" not in sys.modules:
        try:
            import pca  # noqa: F401  (the reference's own pca.py, when the entry script runs from its tree)
        except ImportError:
            from . import pca as _pca
            sys.modules["pca"] = _pca


def install_longvgen_alias(force=False):
    if "longvgen" in sys.modules and not force:
        raise RuntimeError("a `longvgen` package is already imported; refusing to shadow it (pass force=True to override)")
    from . import fifo, pca, pipeline, pipeline_t2to, resampler, scheduler, transformer, vae
    root = types.ModuleType("longvgen")
    root.__path__ = []
    subs = {
        "models": dict(CogVideoXTransformer3DModel=transformer.CogVideoXTransformer3DModel, AutoencoderKLCogVideoX=vae.AutoencoderKLCogVideoX),
        "schedulers": dict(CogVideoXDPMScheduler=scheduler.CogVideoXDPMScheduler),
        "fifo_sampling": dict(cogvideo_fifo_mp_v2=fifo.cogvideo_fifo_mp_v2),
        "pipeline": dict(MPFIFOVideoIPAdapterCogVideoXPipeline=pipeline.MPFIFOVideoIPAdapterCogVideoXPipeline,
                         LongVGenCogVideoXPipeline=pipeline_t2to.LongVGenCogVideoXPipeline),
        "video_ipadapter": dict(Resampler=resampler.Resampler),
    }
    if "pca" not in sys.modules or force:
        sys.modules["pca"] = pca
    sys.modules["longvgen"] = root
    for name, attrs in subs.items():
        m = types.ModuleType(f"longvgen.{name}")
        for k, v in attrs.items():
            setattr(m, k, v)
        setattr(root, name, m)
        sys.modules[f"longvgen.{name}"] = m
    return root
