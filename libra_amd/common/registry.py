"""Model registry — the lookup `train.py` performs to build the model (/root/reference/train.py:29-30:
``registry.get_model_class(cfg.arch).from_config(cfg)``; /root/reference/libra/common/registry.py:58-80, :202-204).

Only the model table is on the hot path's boundary (SURVEY §8b "Registry + factory"); dataset builders, tasks, runners
and lr schedulers of the reference registry belong to its data pipeline / trainer and are out of scope."""


def reference_registry():
    """The reference tree's own registry object (`libra.common.registry.registry`, the one /root/reference/train.py:17 imports)
    when that tree is importable, else None.  `libra` there is a namespace package and the registry module a leaf: importing it
    has no side effects."""
    try:
        from libra.common.registry import registry as ref          # noqa: the REFERENCE's package, not libra_amd.libra
        return ref if hasattr(ref, "mapping") and "model_name_mapping" in ref.mapping else None
    except Exception:
        return None


class Registry:
    mapping = {"model_name_mapping": {}, "state": {}, "paths": {}}

    @classmethod
    def register_model(cls, name):
        """Registers here AND, inside the reference tree, in the reference's registry - so that an unmodified train.py
        (`registry.get_model_class(cfg.arch)`, train.py:29-30, after `from libra.models import *`, :23) builds the MI355X
        wrapper.  There the entry is set directly: the MI355X class replaces whatever the reference registered under the name."""
        def wrap(model_cls):
            if name in cls.mapping["model_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(name, cls.mapping["model_name_mapping"][name]))
            cls.mapping["model_name_mapping"][name] = model_cls
            ref = reference_registry()
            if ref is not None:
                ref.mapping["model_name_mapping"][name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)

    @classmethod
    def list_models(cls):
        return sorted(cls.mapping["model_name_mapping"].keys())


registry = Registry()
