"""Stand-in for the third-party `colorlog` package (not installed here).

TEST INFRASTRUCTURE ONLY.  The reference hot path imports
``colorlog.ColoredFormatter`` (nhd/NHDCommon.py:2, nhd/Matcher.py:7,
nhd/Node.py:6) but only uses it when stderr is a tty (NHDCommon.py:29-32).
"""
import logging


class ColoredFormatter(logging.Formatter):
    def __init__(self, fmt=None, datefmt=None, log_colors=None, **kw):
        super().__init__((fmt or "").replace("%(log_color)s", ""), datefmt)
