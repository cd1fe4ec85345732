from .chamfer_distance import (  # noqa: F401
    ChamferDistance,
    ChamferDistanceFunction,
    ChamferDistanceMean,
    cd,
)
