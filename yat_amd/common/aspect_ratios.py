"""Aspect-ratio bucket tables, carried as data.

[RECALL] diffusers.pipelines.pixart_alpha.pipeline_pixart_alpha.ASPECT_RATIO_{256,512,1024}_BIN and
pipeline_pixart_sigma.ASPECT_RATIO_2048_BIN (imported at /root/reference/train_sana.py:2-3; not in this
container).  Keys are ``str(float ratio)`` with ratio = H / W, values ``[H, W]`` in pixels, exactly the shape
the reference indexes at common/bucket_sampler.py:65,259.  The 1024 table is SURVEY.md App. A.5; the other
resolutions are the same ratios scaled (all entries stay multiples of 32 px at >= 512).  Provisional until it can
be diffed against a diffusers install.
"""

_R1024 = {
    "0.25": [512, 2048], "0.28": [512, 1856], "0.32": [576, 1792], "0.33": [576, 1728], "0.35": [576, 1664],
    "0.4": [640, 1600], "0.42": [640, 1536], "0.48": [704, 1472], "0.5": [704, 1408], "0.52": [704, 1344],
    "0.57": [768, 1344], "0.6": [768, 1280], "0.68": [832, 1216], "0.72": [832, 1152], "0.78": [896, 1152],
    "0.82": [896, 1088], "0.88": [960, 1088], "0.94": [960, 1024], "1.0": [1024, 1024], "1.07": [1024, 960],
    "1.13": [1088, 960], "1.21": [1088, 896], "1.29": [1152, 896], "1.38": [1152, 832], "1.46": [1216, 832],
    "1.67": [1280, 768], "1.75": [1344, 768], "2.0": [1408, 704], "2.09": [1472, 704], "2.4": [1536, 640],
    "2.5": [1600, 640], "3.0": [1728, 576], "4.0": [2048, 512],
}


def _scaled(num, den):
    return {k: [float(h * num // den), float(w * num // den)] for k, (h, w) in _R1024.items()}


ASPECT_RATIO_1024_BIN = {k: [float(h), float(w)] for k, (h, w) in _R1024.items()}
ASPECT_RATIO_512_BIN = _scaled(1, 2)
ASPECT_RATIO_256_BIN = _scaled(1, 4)
ASPECT_RATIO_2048_BIN = _scaled(2, 1)


def table_for_resolution(resolution: int):
    """train_sana.py:45-57: sample_size * 32 -> table."""
    return {256: ASPECT_RATIO_256_BIN, 512: ASPECT_RATIO_512_BIN, 1024: ASPECT_RATIO_1024_BIN}.get(
        resolution, ASPECT_RATIO_2048_BIN)
