"""Name -> class registries with the reference's build-from-config contract.

Mirrors the behaviour callers rely on in openpoints/utils/registry.py:8-294:
``Registry(name)``, ``@R.register_module()`` / ``R.register_module(name=..., module=cls)``,
``R.get(key)``, ``R.build(cfg, **kw)`` and ``build_from_cfg`` -- the config is
deep-copied, ``NAME`` selects the class, every remaining key becomes a
constructor keyword, and a constructor error is re-raised with the class name
in front (registry.py:286-294).
"""
import copy
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'NAME' not in cfg and (default_args is None or 'NAME' not in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "NAME", but got {cfg}\n{default_args}')
    if not isinstance(registry, Registry):
        raise TypeError(f'registry must be a Registry object, but got {type(registry)}')
    if default_args is not None and not isinstance(default_args, dict):
        raise TypeError(f'default_args must be a dict or None, but got {type(default_args)}')

    target = cfg.get('NAME')
    if isinstance(target, str):
        cls = registry.get(target)
        if cls is None:
            raise KeyError(f'{target} is not in the {registry.name} registry')
    elif inspect.isclass(target):
        cls = target
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(target)}')

    try:
        kwargs = copy.deepcopy(cfg)
        if default_args is not None:
            kwargs.update(default_args)
        kwargs.pop('NAME')
        return cls(**kwargs)
    except Exception as e:  # a bare TypeError would not say which class failed
        raise type(e)(f'{cls.__name__}: {e}')


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name = name
        self._module_dict = {}
        self._children = {}
        self._scope = scope if scope is not None else self._caller_package()
        self.parent = parent
        if build_func is not None:
            self.build_func = build_func
        elif parent is not None:
            self.build_func = parent.build_func
        else:
            self.build_func = build_from_cfg
        if parent is not None:
            assert isinstance(parent, Registry)
            parent._add_children(self)

    @staticmethod
    def _caller_package():
        frame = inspect.stack()[2][0]
        mod = inspect.getmodule(frame)
        return mod.__name__.split('.')[0] if mod is not None else '__main__'

    # -- introspection -------------------------------------------------------
    name = property(lambda self: self._name)
    scope = property(lambda self: self._scope)
    module_dict = property(lambda self: self._module_dict)
    children = property(lambda self: self._children)

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={self._module_dict})'

    # -- lookup / build --------------------------------------------------------
    def get(self, key):
        scope, rest = None, key
        dot = key.find('.')
        if dot != -1:
            scope, rest = key[:dot], key[dot + 1:]
        if scope is None or scope == self._scope:
            return self._module_dict.get(rest)
        if scope in self._children:
            return self._children[scope].get(rest)
        root = self
        while root.parent is not None:
            root = root.parent
        return root.get(key) if root is not self else None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _add_children(self, registry):
        assert isinstance(registry, Registry) and registry.scope is not None
        assert registry.scope not in self._children, f'scope {registry.scope} exists in {self.name} registry'
        self._children[registry.scope] = registry

    # -- registration ------------------------------------------------------------
    def _register_module(self, module_class, module_name=None, force=False):
        if not inspect.isclass(module_class):
            raise TypeError(f'module must be a class, but got {type(module_class)}')
        names = module_name if module_name is not None else module_class.__name__
        if isinstance(names, str):
            names = [names]
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._module_dict[n] = module_class

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if isinstance(name, type):  # old style: R.register_module(SomeClass)
            self._register_module(name, force=force)
            return name
        if not (name is None or isinstance(name, str) or
                (isinstance(name, (list, tuple)) and all(isinstance(n, str) for n in name))):
            raise TypeError(f'name must be None, a str or a sequence of str, but got {type(name)}')
        if module is not None:
            self._register_module(module, name, force)
            return module

        def _decorator(cls):
            self._register_module(cls, name, force)
            return cls
        return _decorator
