class UDFException(Exception):
    """Raised when the UDF interface is violated (reference: common/exceptions.py)."""


class ExecutorSpecException(Exception):
    pass


class JobCancelledError(Exception):
    pass


class UDFRunCancelled(Exception):
    pass


class HipRequiredError(RuntimeError):
    """A MI355X-native operator was asked to run without the HIP backend."""
