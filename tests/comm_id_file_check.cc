// Reader side of tr::exchange_comm_id_through_file (include/tauray_hip_comm.hh): which files a rank other than 0 accepts.
// usage: comm_id_file_check <path> <nonce> <timeout seconds>  -> prints "id <first byte>" or "timeout"
#include "tauray_hip_comm.hh"
#include <iostream>
int main(int argc, char** argv)
{
    if(argc < 4) return 2;
    try
    {
        const std::vector<char> id = tr::exchange_comm_id_through_file(argv[1], 1, std::stoull(argv[2]), std::stod(argv[3]), 5.0);
        std::cout << "id " << (int)(unsigned char)id[0] << "\n";
    }
    catch(const std::exception& e) { std::cout << "timeout\n"; }
    return 0;
}
