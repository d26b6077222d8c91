"""Drop-in alias: `slotformer.base_slots` / `slotformer.video_prediction` resolve to the
MI355X-native packages, so `scripts/train.py`-style `importlib.import_module('slotformer.<task>')`
and `from slotformer.base_slots.models import StoSAVi` keep working."""
import importlib
import sys

for _name in ('base_slots', 'video_prediction'):
    _mod = importlib.import_module(f'slotformer_amd.{_name}')
    sys.modules[f'{__name__}.{_name}'] = _mod
    setattr(sys.modules[__name__], _name, _mod)
    sys.modules[f'{__name__}.{_name}.models'] = importlib.import_module(f'slotformer_amd.{_name}.models')
