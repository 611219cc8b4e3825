"""nn.Module base with the reference's verbosity gate and shape-agnostic load_state_dict
(torchpq/CustomModule.py:4-22)."""
import torch.nn as nn


class CustomModule(nn.Module):
    def __init__(self):
        super().__init__()

    def print_message(self, text, min_verbosity=0):
        if getattr(self, "verbose", 0) < min_verbosity:
            return
        print(f"{type(self).__name__}: {text}")

    def load_state_dict(self, state_dict):
        # Buffers change shape as the index grows, so every top-level entry is re-registered
        # rather than copied into the existing tensor; children are handled recursively.
        own = {k: v for k, v in state_dict.items() if "." not in k}
        for key, value in own.items():
            assert hasattr(self, key), f"attribute {key} does not exist"
            delattr(self, key)
            self.register_buffer(key, value)
        for name, child in self.named_children():
            prefix = name + "."
            sub = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
            child.load_state_dict(sub)
        self._after_load_state_dict()

    def _after_load_state_dict(self):
        pass
