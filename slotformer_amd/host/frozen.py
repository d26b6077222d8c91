"""Sub-modules that a model borrows, frozen, from another model's checkpoint (the SlotFormer variants take the SAVi /
STEVE decoder that way: slotformer.py:196-218, steve_slotformer.py:66-84)."""
import torch


def load_prefixed(module, state_dict, prefix):
    """module.load_state_dict(entries of `state_dict` whose key starts with `prefix`, prefix stripped)."""
    n = len(prefix)
    module.load_state_dict({k[n:]: v for k, v in state_dict.items() if k.startswith(prefix)})


def freeze(*modules):
    for m in modules:
        for p in m.parameters():
            p.requires_grad = False
        m.eval()


def checkpoint_state(path, what):
    assert path, f'Please provide pretrained {what} weight'
    return torch.load(path, map_location='cpu')['state_dict']
