"""slotformer.base_slots counterpart: exports build_model (build_dataset/build_method are the
reference's host-side data/training orchestration -- out of scope)."""
from .models import build_model


def build_dataset(params, *a, **k):
    raise NotImplementedError('datasets are host-side I/O outside the hot path (SURVEY.md 2.1 row 13)')


def build_method(*a, **k):
    raise NotImplementedError('the nerv trainer is outside the hot path (SURVEY.md 2.1 row 12)')
