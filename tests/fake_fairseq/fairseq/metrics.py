LOG = {}


def log_scalar(key, value, weight=1, priority=10, round=None):
    LOG[key] = float(value)
