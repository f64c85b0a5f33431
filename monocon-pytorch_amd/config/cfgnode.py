"""Minimal yacs-compatible ``CfgNode`` (yacs is not installed in the MI355X image).

Covers the subset the reference uses (config/monocon_configs.py, utils/engine_utils.py:38-52,
engine/*.py): attribute access on nested nodes, ``clone``, ``get``, ``set_new_allowed``,
``merge_from_file`` (YAML), ``dump``.  When the real ``yacs`` is importable it is used instead.
"""
import copy

import yaml

try:                                            # pragma: no cover - not installed here
    from yacs.config import CfgNode             # noqa: F401
except Exception:

    class CfgNode(dict):
        def __init__(self, init=None, new_allowed=False):
            super().__init__()
            object.__setattr__(self, "_new_allowed", new_allowed)
            for k, v in (init or {}).items():
                self[k] = CfgNode(v, new_allowed) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            return copy.deepcopy(self)

        def set_new_allowed(self, flag):
            object.__setattr__(self, "_new_allowed", flag)
            for v in self.values():
                if isinstance(v, CfgNode):
                    v.set_new_allowed(flag)

        def _merge(self, other, path=""):
            for k, v in other.items():
                if k not in self and not self._new_allowed:
                    raise KeyError("Non-existent config key: %s%s" % (path, k))
                if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                    self[k]._merge(v, path + k + ".")
                else:
                    self[k] = CfgNode(v, self._new_allowed) if isinstance(v, dict) else v

        def merge_from_file(self, path):
            with open(path) as f:
                self._merge(yaml.safe_load(f) or {})

        def merge_from_list(self, kv):
            for k, v in zip(kv[0::2], kv[1::2]):
                node = self
                parts = k.split(".")
                for p in parts[:-1]:
                    node = node[p]
                node[parts[-1]] = v

        def _plain(self):
            return {k: (v._plain() if isinstance(v, CfgNode) else v) for k, v in self.items()}

        def dump(self, **kw):
            return yaml.safe_dump(self._plain(), **kw)
