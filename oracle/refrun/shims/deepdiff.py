"""Empty stand-in: the reference imports `from deepdiff import DeepDiff` (executor.py:19) and never
uses it.  TEST INFRASTRUCTURE ONLY."""


class DeepDiff(object):
    pass
