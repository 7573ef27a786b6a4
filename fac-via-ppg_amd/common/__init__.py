"""Drop-in for the hot-path half of the reference's ``common`` package (src/common).

Unlike the reference's ``common/__init__.py:1-4`` this does not import the Kaldi/TextGrid
corpus tooling (decode, feat, align, utterance): PPG extraction is upstream of the synthesis
hot path and out of scope."""
