"""Name -> class registry with the fvcore.common.registry.Registry surface the
reference uses (`@REG.register()`, `REG.get(name)`), so `modules.build` /
`model.build` keep their shape without the fvcore dependency."""


class Registry:
    def __init__(self, name):
        self._name = name
        self._table = {}

    def _add(self, key, obj):
        if key in self._table:
            raise KeyError(f"'{key}' is already registered in the '{self._name}' registry")
        self._table[key] = obj

    def register(self, obj=None):
        if obj is None:                       # used as @REG.register()
            def deco(cls_or_fn):
                self._add(cls_or_fn.__name__, cls_or_fn)
                return cls_or_fn
            return deco
        self._add(obj.__name__, obj)          # used as REG.register(obj)
        return obj

    def get(self, name):
        try:
            return self._table[name]
        except KeyError:
            raise KeyError(f"no object named '{name}' in the '{self._name}' registry") from None

    def __contains__(self, name):
        return name in self._table

    def __iter__(self):
        return iter(self._table.items())
