"""Host-side data semantics of the hot path (id assignment, batching order, filter sets) and the
synthetic benchmark-shaped graphs used by bench.py (the reference's datasets are downloads)."""
from .synthetic import SYNTH_SHAPES, make_synthetic_kg  # noqa: F401
from .loaders import add_reciprocal_relations, load_from_csv  # noqa: F401,E402
