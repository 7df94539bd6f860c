"""The names user code imports from the reference's `libertem.common.executor` (common/executor.py:41-175) that
exist in this package: the executor base class, the worker environment and the errors a run can end with."""
from libertem_amd.executor.base import JobExecutor, Environment                       # noqa: F401
from libertem_amd.common.exceptions import JobCancelledError, ExecutorSpecException  # noqa: F401


class ExecutorError(Exception):
    """base class of executor-side failures (common/executor.py:41-42)"""
