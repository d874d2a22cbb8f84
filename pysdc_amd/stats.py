"""Statistics with the reference's shape: a dict ``{Entry(...): value}`` (pySDC/core/hooks.py:9-19,52-66) and
the sorting / filtering helpers of pySDC/helpers/stats_helper.py:4-111."""
from collections import namedtuple

Entry = namedtuple('Entry', ['process', 'process_sweeper', 'time', 'level', 'iter', 'sweep', 'type', 'num_restarts'])


def filter_stats(stats, comm=None, recomputed=None, **kwargs):
    result = {}
    for k, v in stats.items():
        if all(getattr(k, key) == val for key, val in kwargs.items()):
            result[k] = v
    return result


def sort_stats(stats, sortby='time', comm=None):
    return sorted([(getattr(k, sortby), v) for k, v in stats.items()], key=lambda x: x[0])


def get_sorted(stats, sortby='time', comm=None, **kwargs):
    return sort_stats(filter_stats(stats, **kwargs), sortby=sortby)


def get_list_of_types(stats):
    return sorted({k.type for k in stats})
