"""Input pipeline of the MI355X build: mirror of the reference's ``tools/`` package (`tools/dataset.py`)."""
