"""Host-side helpers shared by the reference-shaped model classes: a table-driven `build_model`, loss values computed on
device tensors, and loading of frozen sub-modules from another model's checkpoint.  Nothing here does model arithmetic
on the hot path -- that lives in libslotformer_hip (see `engine.py`)."""
