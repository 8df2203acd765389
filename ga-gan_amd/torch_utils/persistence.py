"""Constructor arguments kept with the object.

The reference decorates its network classes with `persistence.persistent_class` (torch_utils/persistence.py:33-130): a
pickle of such an object carries the defining module's source text, and the object remembers its constructor arguments
(`init_args`, `init_kwargs`).  Nothing here embeds or runs source.  `records_init` gives a class the second half only:
what a snapshot writer (`legacy.save_network_pkl`) and the reference's "rebuild from `init_kwargs`, then copy the
tensors" idiom (persistence.py:86-89, legacy.py:49-60) need.
"""
import copy
import functools

_recording = set()


def records_init(cls):
    """Class decorator: an instance keeps deep copies of the arguments it was built with as `_init_args` /
    `_init_kwargs` (the attribute names the reference's records hold) and hands out copies through `init_args` /
    `init_kwargs`."""
    if cls in _recording:
        return cls
    plain_init = cls.__init__

    @functools.wraps(plain_init)
    def __init__(self, *args, **kwargs):
        plain_init(self, *args, **kwargs)
        self._init_args = copy.deepcopy(args)
        self._init_kwargs = copy.deepcopy(kwargs)

    cls.__init__ = __init__
    cls.init_args = property(lambda self: copy.deepcopy(self._init_args))
    cls.init_kwargs = property(lambda self: copy.deepcopy(self._init_kwargs))
    _recording.add(cls)
    return cls


def records_its_init(obj):
    """Whether `obj` (an instance or a class) was declared with `records_init`."""
    return (obj if isinstance(obj, type) else type(obj)) in _recording
