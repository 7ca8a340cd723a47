"""The reference's scripts ask for an AzureML run handle before training (train_confignet.py:31-32); there is no AzureML
here: the handle is None and logging to it is a no-op (what the reference does off-cluster, azure_ml_utils.py)."""


def get_aml_run():
    return None


def log_job_params(aml_run, args):
    return None
