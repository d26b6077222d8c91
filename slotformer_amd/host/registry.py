"""Table-driven model factory.

The reference's two `build_model(params)` functions (base_slots/models/__init__.py:9-34, video_prediction/models/
__init__.py:6-36) dispatch on `params.model` and forward a fixed set of `params` attributes as keyword arguments.  Here
that mapping is data: model name -> (class, {constructor argument: params attribute})."""


class ModelTable:

    def __init__(self):
        self._rows = {}

    def add(self, name, cls, **arg_from_attr):
        """constructor argument name = params attribute name"""
        self._rows[name] = (cls, arg_from_attr)
        return self

    def build(self, params):
        row = self._rows.get(params.model)
        if row is None:
            raise NotImplementedError(f'{params.model} is not implemented.')
        cls, mapping = row
        return cls(**{arg: getattr(params, attr) for arg, attr in mapping.items()})


# argument groups shared by several models
SLOT_MODEL_ARGS = dict(resolution='resolution', clip_len='input_frames', slot_dict='slot_dict', enc_dict='enc_dict',
                       dec_dict='dec_dict', pred_dict='pred_dict', loss_dict='loss_dict')
ROLLOUT_MODEL_ARGS = dict(resolution='resolution', clip_len='input_frames', slot_dict='slot_dict', dec_dict='dec_dict',
                          rollout_dict='rollout_dict', loss_dict='loss_dict')
