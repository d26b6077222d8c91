"""nerv.training: the two base classes the model / config files subclass."""
from slotformer_amd.nerv_compat import BaseModel, BaseParams  # noqa: F401

__all__ = ['BaseModel', 'BaseParams']


def __getattr__(name):
    if name in ('BaseMethod', 'BaseDataModule', 'CosineAnnealingWarmupRestarts'):
        raise ImportError(f'nerv.training.{name}: the nerv trainer / data modules are outside the hot path this repository '
                          'implements (SURVEY.md 2.1 rows 12-13); install the real nerv package for them')
    raise AttributeError(name)
