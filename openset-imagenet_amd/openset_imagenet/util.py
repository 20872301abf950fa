"""Configuration container with the semantics of the reference's util.NameSpace / util.load_yaml
(openset_imagenet/util.py:16-34): attribute access over a nested YAML mapping, `dump()` back to YAML text.
Only the config contract is mirrored here; OSCR / plotting helpers of the reference are out of the hot path.
"""
import yaml


class NameSpace:
    def __init__(self, config):
        self.update(config)

    def update(self, config):
        for key, value in config.items():
            setattr(self, key, NameSpace(value) if isinstance(value, dict) else value)

    def dict(self):
        return {k: (v.dict() if isinstance(v, NameSpace) else v) for k, v in vars(self).items()}

    def dump(self, indent=4):
        return yaml.dump(self.dict(), indent=indent)

    def __repr__(self):
        return "NameSpace(" + repr(self.dict()) + ")"


def load_yaml(yaml_file):
    """Load a YAML configuration file into a NameSpace."""
    with open(yaml_file, "r") as handle:
        return NameSpace(yaml.safe_load(handle))
