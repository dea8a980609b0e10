from .rand_tps import RandTPS  # noqa: F401
