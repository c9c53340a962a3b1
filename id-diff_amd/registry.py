"""One name -> class registry used for score models, evaluation modules and data modules.

The reference keeps three hand-written copies of the same decorator (models/utils.py:24-47,
lightning_modules/utils.py:1-22, lightning_data_modules/utils.py:4-26); here the behaviour -- usable bare or with
``name=``, duplicate names rejected -- lives once."""


class Registry:
    def __init__(self, kind):
        self.kind = kind
        self._classes = {}

    def register(self, cls=None, *, name=None):
        """``@registry.register`` or ``@registry.register(name='x')``."""
        def bind(klass):
            key = klass.__name__ if name is None else name
            if key in self._classes:
                raise ValueError(f'Already registered model with name: {key}')
            self._classes[key] = klass
            return klass

        return bind if cls is None else bind(cls)

    def get(self, name):
        try:
            return self._classes[name]
        except KeyError:
            raise KeyError(f"unknown {self.kind} {name!r}; registered: {sorted(self._classes)}") from None

    def names(self):
        return sorted(self._classes)
