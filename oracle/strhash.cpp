// TEST TOOL: prints std::hash<std::string>(argv[1]) -- the value the reference uses to name index files
// (/root/reference/src/sortmerna/util.cpp:216-222, index.cpp:75-77), so tests can hand OUR index files to sortmerna_ref.
#include <functional>
#include <iostream>
#include <string>
int main(int argc, char** argv) {
  if (argc < 2) return 1;
  std::cout << std::hash<std::string>{}(std::string(argv[1])) << std::endl;
  return 0;
}
