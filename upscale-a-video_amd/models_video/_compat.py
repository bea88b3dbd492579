"""Minimal, dependency-free equivalents of the diffusers glue the reference classes inherit from
(ConfigMixin / register_to_config / ModelMixin / BaseOutput; reference imports at
models_video/unet_video.py:24-33).  Behaviour kept: `X.from_config(path_or_dict)` filters keys by
the constructor signature and drops `_`-prefixed ones; `.config` is an attribute dict of the
bound constructor arguments including defaults.
"""
import inspect
import json
import os
from collections import OrderedDict
from dataclasses import fields

import torch
import torch.nn as nn


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        if "_internal_dict" not in self.__dict__:
            object.__setattr__(self, "_internal_dict", FrozenDict())
        self.__dict__["_internal_dict"].update(kwargs)

    @property
    def config(self):
        return self.__dict__["_internal_dict"]

    @classmethod
    def load_config(cls, path):
        if os.path.isdir(path):
            path = os.path.join(path, cls.config_name)
        with open(path) as fh:
            return json.load(fh)

    @classmethod
    def from_config(cls, config, **kwargs):
        if isinstance(config, (str, os.PathLike)):
            config = cls.load_config(config)
        params = inspect.signature(cls.__init__).parameters
        init = {k: v for k, v in dict(config).items() if k in params and not k.startswith("_")}
        init.update({k: v for k, v in kwargs.items() if k in params})
        return cls(**init)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapper(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self" and not k.startswith("_")}
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    wrapper.__signature__ = sig
    wrapper.__wrapped__ = init
    return wrapper


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        for p in self.parameters():
            if p.is_floating_point():
                return p.dtype
        return torch.float32

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(OrderedDict):
    """Ordered dict with attribute and index access, populated from dataclass fields."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def __setattr__(self, name, value):
        if name in self.keys() and value is not None:
            super().__setitem__(name, value)
        super().__setattr__(name, value)

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())
