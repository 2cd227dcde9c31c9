"""Minimal stand-in for the `omegaconf` package (not installed in this image), used ONLY by
tests/golden/make_golden.py to import the reference's model code.  Covers the API surface the
reference touches: DictConfig attribute/item access, OmegaConf.create / to_container."""


class DictConfig(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return DictConfig({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [_wrap(v) for v in obj]
    return obj


def _unwrap(obj):
    if isinstance(obj, dict):
        return {k: _unwrap(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_unwrap(v) for v in obj]
    return obj


class OmegaConf:
    @staticmethod
    def create(obj=None):
        return _wrap(obj if obj is not None else {})

    @staticmethod
    def to_container(cfg, resolve=False):
        return _unwrap(cfg)
