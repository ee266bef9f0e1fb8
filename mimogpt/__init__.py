"""Import-path shim: `mimogpt.infer.SelftokPipeline` resolves to the MI355X-native implementation in
selftoktokenizer_amd (nothing from the reference tree lives here)."""
