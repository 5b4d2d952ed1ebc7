"""Attribute-access config dict with YAML loading and CLI overrides.

Behavioural mirror of openpoints/utils/config.py:19-99 (``EasyConfig``): nested
dicts become EasyConfig on ``update``, ``load(path, recursive=True)`` layers
every ``default.yaml`` found from the filesystem root down to the file, and a
list/tuple passed to ``update`` is parsed as ``key=value`` / ``--key value``
dotted overrides with ``literal_eval`` values.  (The reference overloads
``update`` with the third-party ``multimethod`` package; a type test does the
same job here.)
"""
import hashlib
import json
import os
from ast import literal_eval

import yaml


class EasyConfig(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def load(self, fpath, *, recursive=False):
        if not os.path.exists(fpath):
            raise FileNotFoundError(fpath)
        chain = [fpath]
        if recursive:
            ext = os.path.splitext(fpath)[1]
            d = fpath
            while os.path.dirname(d) != d:
                d = os.path.dirname(d)
                chain.append(os.path.join(d, 'default' + ext))
        for path in reversed(chain):
            if os.path.exists(path):
                with open(path) as f:
                    self.update(yaml.safe_load(f))

    def reload(self, fpath, *, recursive=False):
        self.clear()
        self.load(fpath, recursive=recursive)

    def update(self, other=None, **kw):
        if isinstance(other, (list, tuple)):
            self._update_from_opts(other)
        elif other is not None:
            self._update_from_dict(other)
        if kw:
            self._update_from_dict(kw)

    def _update_from_dict(self, other):
        for key, value in other.items():
            if isinstance(value, dict):
                if key not in self or not isinstance(self[key], EasyConfig):
                    self[key] = EasyConfig()
                self[key].update(value)
            else:
                self[key] = value

    def _update_from_opts(self, opts):
        i = 0
        while i < len(opts):
            opt = opts[i]
            if opt.startswith('--'):
                opt = opt[2:]
            if '=' in opt:
                key, value = opt.split('=', 1)
                i += 1
            else:
                key, value = opt, opts[i + 1]
                i += 2
            try:
                value = literal_eval(value)
            except Exception:
                pass
            node = self
            *parents, leaf = key.split('.')
            for p in parents:
                node = node.setdefault(p, EasyConfig())
            node[leaf] = value

    def dict(self):
        return {k: (v.dict() if isinstance(v, EasyConfig) else v) for k, v in self.items()}

    def hash(self):
        return hashlib.sha256(json.dumps(self.dict(), sort_keys=True).encode()).hexdigest()
