"""boto3 is an import-time dependency of BERT/file_utils.py:32 (S3 model download, not on the train step)."""


def resource(*a, **k):
    raise RuntimeError("boto3 shim: S3 access is not available")


client = resource
