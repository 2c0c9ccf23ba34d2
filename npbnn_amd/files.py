"""Persistence helpers (reference: np_bnn/BNN_files.py:260-267, BNN_lib.py:241-243)."""
import pickle


def load_obj(file_name):
    with open(file_name, 'rb') as f:
        return pickle.load(f)


def SaveObject(obj, filename):
    with open(filename, 'wb') as output:
        pickle.dump(obj, output, pickle.HIGHEST_PROTOCOL)
