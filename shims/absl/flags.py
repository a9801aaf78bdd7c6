import sys


class _Flag:
    def __init__(self, name, default, help, parser, choices=None):
        self.name, self.default, self.value, self.help, self.parser, self.choices = name, default, default, help, parser, choices

    def parse(self, text):
        v = self.parser(text)
        if self.choices is not None and v not in self.choices:
            raise ValueError("flag --%s=%r must be one of %s" % (self.name, v, self.choices))
        self.value = v


class FlagValues:
    def __init__(self):
        object.__setattr__(self, "_flags", {})
        object.__setattr__(self, "_parsed", False)

    def _define(self, flag):
        self._flags[flag.name] = flag

    def __getattr__(self, name):
        flags = object.__getattribute__(self, "_flags")
        if name in flags:
            return flags[name].value
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._flags[name].value = value

    def __getitem__(self, name):
        return self._flags[name]

    def __contains__(self, name):
        return name in self._flags

    def set_default(self, name, value):
        f = self._flags[name]
        if f.value == f.default:
            f.value = value
        f.default = value

    def flag_values_dict(self):
        return {k: f.value for k, f in self._flags.items()}

    def __call__(self, argv, known_only=False):
        rest = [argv[0]] if argv else []
        it = iter(argv[1:])
        for a in it:
            if not a.startswith("--"):
                rest.append(a)
                continue
            body = a[2:]
            if "=" in body:
                name, text = body.split("=", 1)
            else:
                name, text = body, None
            if name.startswith("no") and name[2:] in self._flags and text is None and self._flags[name[2:]].parser is _bool:
                self._flags[name[2:]].value = False
                continue
            if name not in self._flags:
                if known_only:
                    rest.append(a)
                    continue
                raise ValueError("unknown flag --%s" % name)
            f = self._flags[name]
            if text is None:
                if f.parser is _bool:
                    f.value = True
                    continue
                text = next(it)
            f.parse(text)
        object.__setattr__(self, "_parsed", True)
        return rest


def _bool(text):
    if isinstance(text, bool):
        return text
    t = str(text).lower()
    if t in ("1", "true", "t", "yes", "y"):
        return True
    if t in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError("not a boolean: %r" % text)


def _list(text):
    if isinstance(text, (list, tuple)):
        return list(text)
    return [s for s in str(text).split(",") if s != ""]


FLAGS = FlagValues()


def DEFINE_string(name, default, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, default, help, lambda s: None if s is None else str(s)))


def DEFINE_integer(name, default, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, default, help, int))


def DEFINE_float(name, default, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, default, help, float))


def DEFINE_boolean(name, default, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, default, help, _bool))


DEFINE_bool = DEFINE_boolean


def DEFINE_enum(name, default, enum_values, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, default, help, str, choices=list(enum_values)))


def DEFINE_list(name, default, help=None, flag_values=FLAGS, **kw):
    flag_values._define(_Flag(name, _list(default) if default is not None else None, help, _list))


def mark_flag_as_required(name, flag_values=FLAGS):
    pass


_ = sys
