import logging as _logging

DEBUG, INFO, WARNING, ERROR, FATAL = _logging.DEBUG, _logging.INFO, _logging.WARNING, _logging.ERROR, _logging.CRITICAL
_log = _logging.getLogger("absl")
_seen = {}


def log(level, msg, *args):
    _log.log(level, msg, *args)


def info(msg, *args):
    _log.info(msg, *args)


def warning(msg, *args):
    _log.warning(msg, *args)


warn = warning


def error(msg, *args):
    _log.error(msg, *args)


def log_first_n(level, msg, n, *args):
    c = _seen.get(msg, 0)
    if c < n:
        _seen[msg] = c + 1
        _log.log(level, msg, *args)


def set_verbosity(v):
    pass
