"""Exception types raised on this path, under the names pySDC users catch (pySDC/core/errors.py).

``EngineError`` is this package's own: the HIP library is missing or a C-ABI call failed (there is no CPU
fallback)."""


def _exception(name, doc, base=Exception):
    return type(name, (base,), {'__doc__': doc, '__module__': __name__})


DataError = _exception('DataError', 'a datatype was used with incompatible data')
ParameterError = _exception('ParameterError', 'a parameter dictionary is incomplete or inconsistent')
UnlockError = _exception('UnlockError', 'a level was used before a predictor / restriction unlocked it')
CollocationError = _exception('CollocationError', 'the collocation rule could not be built')
ConvergenceError = _exception('ConvergenceError', 'an iteration failed to converge')
TransferError = _exception('TransferError', 'space or node transfer between levels is not possible')
CommunicationError = _exception('CommunicationError', 'a message between time steps does not match its tag')
ControllerError = _exception('ControllerError', 'the controller reached an inconsistent state')
ProblemError = _exception('ProblemError', 'a problem class rejected its parameters or its solver failed')
EngineError = _exception('EngineError', 'libsdcmi.so is missing or a C-ABI call failed', RuntimeError)


class ReadOnlyError(Exception):
    """assignment to a registered read-only problem attribute"""

    def __init__(self, name):
        super().__init__(f'cannot set read-only attribute {name}')
