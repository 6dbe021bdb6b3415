"""MI355X-native Goldilocks IBDWT squaring engine (the Marin hot path of cherubrock-seb/PrMers).

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI -> libmi355_engine.so) and the
host-side mirror of the reference's `engine` interface (engine.py) plus its PRP / LL callers (prp.py).
"""
from .engine import CrtEngine, Engine, EngineError, load_library, resolve_plan, LIB_PATH  # noqa: F401
