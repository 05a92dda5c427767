"""Model registry — the lookup `train.py` performs to build the model (/root/reference/train.py:29-30:
``registry.get_model_class(cfg.arch).from_config(cfg)``; /root/reference/libra/common/registry.py:58-80, :202-204).

Only the model table is on the hot path's boundary (SURVEY §8b "Registry + factory"); dataset builders, tasks, runners
and lr schedulers of the reference registry belong to its data pipeline / trainer and are out of scope."""


class Registry:
    mapping = {"model_name_mapping": {}, "state": {}, "paths": {}}

    @classmethod
    def register_model(cls, name):
        def wrap(model_cls):
            if name in cls.mapping["model_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(name, cls.mapping["model_name_mapping"][name]))
            cls.mapping["model_name_mapping"][name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)

    @classmethod
    def list_models(cls):
        return sorted(cls.mapping["model_name_mapping"].keys())


registry = Registry()
