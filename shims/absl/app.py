import sys

from . import flags


class UsageError(Exception):
    pass


def run(main, argv=None):
    rest = flags.FLAGS(sys.argv if argv is None else argv)
    sys.exit(main(rest))
