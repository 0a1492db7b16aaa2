"""Exception types of the public API (same names and meaning as bayesloop/exceptions.py:7-16 in the reference)."""


class ConfigurationError(Exception):
    """Some part of the configuration of a study is inconsistent (raised before any device work starts)."""


class PostProcessingError(Exception):
    """A post-processing accessor was called with arguments that do not match the fitted study."""


class BackendError(RuntimeError):
    """libblhip.so is missing, not loadable, or no MI355X is visible.  There is no CPU fallback."""
