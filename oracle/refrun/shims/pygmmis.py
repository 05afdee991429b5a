"""Empty stand-in: the reference imports `from pygmmis import GMM` (traceweaver_v3.py:20) and never
uses it.  TEST INFRASTRUCTURE ONLY."""


class GMM(object):
    pass
