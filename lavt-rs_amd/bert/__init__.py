"""Drop-in for the reference's `./bert` directory (a copy of HF transformers 3.0.2 that the reference imports as
`from bert.modeling_bert import BertModel`; lib/_utils.py:7, train.py:12, test.py:17) -- the text encoder on liblavt_hip."""
