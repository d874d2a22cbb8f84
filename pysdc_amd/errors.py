"""Exception types with the names the reference raises on this path
(/root/reference/pySDC/core/errors.py:1-90), so callers can catch the same classes."""


class DataError(Exception):
    pass


class ParameterError(Exception):
    pass


class UnlockError(Exception):
    pass


class CollocationError(Exception):
    pass


class ConvergenceError(Exception):
    pass


class TransferError(Exception):
    pass


class CommunicationError(Exception):
    pass


class ControllerError(Exception):
    pass


class ProblemError(Exception):
    pass


class ReadOnlyError(Exception):
    def __init__(self, name):
        super().__init__(f'cannot set read-only attribute {name}')


class EngineError(RuntimeError):
    """Raised when the HIP engine library is missing or a C-ABI call fails."""
