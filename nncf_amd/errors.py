"""Exception types with the names and meaning of src/nncf/errors.py:13-142 (the subset the path raises)."""


class InternalError(Exception):
    pass


class ValidationError(Exception):
    pass


class UnsupportedModelError(Exception):
    pass


class UnsupportedBackendError(Exception):
    pass


class ParameterNotSupportedError(Exception):
    pass


class InvalidGroupSizeError(Exception):
    pass
