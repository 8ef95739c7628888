// placeholder until the driver lands (next commit)
int main() { return 0; }
