import datetime


class _Delta:
    def __init__(self, sec):
        self.sec = sec


class Time:  # import-only stand-in (time_logger.py only needs now() and subtraction)
    def __init__(self, val=None, format=None, **kw):
        self.val = val

    @staticmethod
    def now():
        return Time(datetime.datetime.now())

    def __sub__(self, other):
        return _Delta((self.val - other.val).total_seconds())


class TimeDelta:
    def __init__(self, val=None, format=None, **kw):
        self.val = val
