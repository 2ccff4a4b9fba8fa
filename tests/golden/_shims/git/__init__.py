"""Scratch shim used ONLY by tests/golden/make_golden.py (GitPython is not installed).
The reference only asks "is this a git repo?" for metadata; answer "no"."""


class exc:
    class InvalidGitRepositoryError(Exception):
        pass


class Repo:
    def __init__(self, *a, **kw):
        raise exc.InvalidGitRepositoryError()
