class CudaSupportError(Exception):
    pass


gpus = []


def is_available():
    return False
