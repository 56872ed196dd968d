"""Name -> object registries (basedet/utils/registry.py:12-26 keeps `registers.models`, `registers.solvers`, ...)."""


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, name=None):
        def deco(obj):
            self[name or obj.__name__] = obj
            return obj
        return deco

    def get(self, name):  # noqa: A003 - mirrors basecore Registry.get (raises on a missing name)
        if name not in self:
            raise KeyError(f"{name!r} is not registered in {self._name}")
        return self[name]


class registers:  # noqa: N801 - reference spelling
    models = Registry("models")
    solvers = Registry("solvers")
    hooks = Registry("hooks")
    trainers = Registry("trainers")
