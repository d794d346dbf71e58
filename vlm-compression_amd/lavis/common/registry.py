"""Pruner table of `lavis.common.registry` (reference: lavis/common/registry.py:112-139,
269-270).  The reference class also registers models, tasks, processors, runners ...;
those tables are outside the compression hot path and live on in the reference's own
file -- when integrating, keep the reference registry and nothing here is needed."""


class Registry:
    mapping = {"pruner_name_mapping": {}, "task_name_mapping": {}, "state": {}, "paths": {}}

    @classmethod
    def register_pruner(cls, name):
        def wrap(pruner_cls):
            from lavis.compression.pruners.base_pruner import BasePruner
            assert issubclass(pruner_cls, BasePruner), "All pruners must inherit BasePruner class"
            if name in cls.mapping["pruner_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(
                    name, cls.mapping["pruner_name_mapping"][name]))
            cls.mapping["pruner_name_mapping"][name] = pruner_cls
            return pruner_cls
        return wrap

    @classmethod
    def register_task(cls, name):
        """Same contract as the reference's `register_task` (lavis/common/registry.py:52-76)."""
        def wrap(task_cls):
            if name in cls.mapping["task_name_mapping"]:
                raise KeyError("Name '{}' already registered for {}.".format(name, cls.mapping["task_name_mapping"][name]))
            cls.mapping["task_name_mapping"][name] = task_cls
            return task_cls
        return wrap

    @classmethod
    def get_task_class(cls, name):
        return cls.mapping["task_name_mapping"].get(name, None)

    @classmethod
    def get_pruner_class(cls, name):
        return cls.mapping["pruner_name_mapping"].get(name, None)

    @classmethod
    def list_pruners(cls):
        return sorted(cls.mapping["pruner_name_mapping"].keys())


registry = Registry()
