"""Minimal dllogger-compatible writer (the reference logs through NVIDIA's dllogger, un-vendored: SURVEY.md 8b).

Same call surface as the subset the hot-path recipes use (init / log / metadata / flush, StdOutBackend,
JSONStreamBackend, Verbosity) and the same on-disk format: one `DLLL {json}` line per record with the keys
"timestamp", "datetime", "elapsedtime", "type", "step", "data" (+ "metadata" records).
"""
import atexit
import json
import sys
import time
from collections import defaultdict
from datetime import datetime


class Verbosity:
    OFF = -1
    DEFAULT = 0
    VERBOSE = 1


class Backend:
    def __init__(self, verbosity):
        self._verbosity = verbosity

    @property
    def verbosity(self):
        return self._verbosity


class JSONStreamBackend(Backend):
    def __init__(self, verbosity, filename, append=False):
        super().__init__(verbosity)
        self.file = open(filename, "a" if append else "w")
        atexit.register(self.file.close)

    def _write(self, rec):
        self.file.write("DLLL " + json.dumps(rec) + "\n")

    def metadata(self, timestamp, elapsedtime, metric, metadata):
        self._write(dict(timestamp=str(timestamp.timestamp()), elapsedtime=str(elapsedtime), datetime=str(timestamp),
                         type="METADATA", metric=metric, metadata=metadata))

    def log(self, timestamp, elapsedtime, step, data):
        self._write(dict(timestamp=str(timestamp.timestamp()), datetime=str(timestamp), elapsedtime=str(elapsedtime),
                         type="LOG", step=step, data=data))

    def flush(self):
        self.file.flush()


class StdOutBackend(Backend):
    def __init__(self, verbosity, step_format=None, metric_format=None, prefix_format=None):
        super().__init__(verbosity)
        self._metadata = defaultdict(dict)
        self.step_format = step_format or (lambda step: str(step))

    def metadata(self, timestamp, elapsedtime, metric, metadata):
        self._metadata[metric].update(metadata)

    def log(self, timestamp, elapsedtime, step, data):
        body = " ".join("%s : %s" % (k, ("%.6g" % v) if isinstance(v, float) else v) for k, v in data.items())
        print("DLL %s - %s %s" % (timestamp, self.step_format(step), body))

    def flush(self):
        sys.stdout.flush()


class _Logger:
    def __init__(self):
        self.backends, self.t0 = [], time.time()

    def init(self, backends):
        self.backends, self.t0 = list(backends), time.time()

    def metadata(self, metric, metadata):
        now = datetime.now()
        for b in self.backends:
            b.metadata(now, time.time() - self.t0, metric, metadata)

    def log(self, step, data, verbosity=Verbosity.DEFAULT):
        now = datetime.now()
        for b in self.backends:
            if b.verbosity >= verbosity:
                b.log(now, time.time() - self.t0, step, data)

    def flush(self):
        for b in self.backends:
            b.flush()


_GLOBAL = _Logger()
init, metadata, log, flush = _GLOBAL.init, _GLOBAL.metadata, _GLOBAL.log, _GLOBAL.flush
